"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY UNPINNED: the reference holds no golden vectors (see sph_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsph_oracle.so")

WCSPH, WCXSPH, WCXSPH3D = 0, 1, 2
STATE_N, STATE_NP1 = 0, 1
SEARCH_BRUTE, SEARCH_CELLS = 0, 1


class Case(C.Structure):
    _fields_ = [("variant", C.c_int32), ("npart", C.c_int32 * 3), ("pstep", C.c_double),
                ("box", C.c_double * 3), ("start", C.c_double * 3), ("dt0", C.c_double),
                ("simtype", C.c_int32), ("reserved_", C.c_int32), ("drop_radius", C.c_double),
                ("drop_centre", C.c_double * 3), ("drop_height", C.c_double), ("drop_v", C.c_double * 3),
                ("rho0", C.c_double), ("Cs", C.c_double), ("alpha", C.c_double), ("gravity", C.c_double)]


class Consts(C.Structure):
    _fields_ = [("dim", C.c_int32), ("sim_pts", C.c_uint64), ("bound_parts", C.c_uint64),
                ("npts", C.c_uint64), ("simmass", C.c_double), ("boundmass", C.c_double),
                ("rho0", C.c_double), ("Cs", C.c_double), ("gam", C.c_double), ("B", C.c_double),
                ("H", C.c_double), ("HSQ", C.c_double), ("correc", C.c_double),
                ("search_radius", C.c_double), ("alpha", C.c_double), ("eps", C.c_double),
                ("beta", C.c_double), ("subits", C.c_int32), ("g_axis", C.c_int32),
                ("g", C.c_double)]


def build(force=False):
    """Compile the oracle with its Makefile (g++, reference flags)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "sph_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Case)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_default_case.argtypes = [C.c_int32, C.POINTER(Case)]
        L.orc_derive.argtypes = [C.POINTER(Case), C.POINTER(Consts)]
        L.orc_derive.restype = C.c_int
        L.orc_get_consts.argtypes = [C.c_void_p, C.POINTER(Consts)]
        L.orc_set_search.argtypes = [C.c_void_p, C.c_int32]
        vp = C.c_void_p
        L.orc_get_state.argtypes = [vp, C.c_int32] + [vp] * 8
        L.orc_set_state.argtypes = [vp, C.c_int32] + [vp] * 8
        L.orc_forces.argtypes = [vp, C.c_int32]
        L.orc_zero_boundary_f.argtypes = [vp, C.c_int32]
        L.orc_step.argtypes = [vp, C.c_int32, C.c_int32]
        L.orc_density_reinit.argtypes = [vp]
        L.orc_get_scalars.argtypes = [vp] + [vp] * 6
        L.orc_set_dt.argtypes = [vp, C.c_double]
        L.orc_neighbours.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
        L.orc_w2kernel.argtypes = [vp, C.c_double]
        L.orc_w2kernel.restype = C.c_double
        L.orc_w2gradk.argtypes = [vp, vp, C.c_double, vp]
        _lib = L
    return _lib


def default_case(variant):
    c = Case()
    lib().orc_default_case(variant, C.byref(c))
    return c


def make_case(variant, npart, pstep, box, start, dt0=0.0002, drop=None, phys=None):
    """drop: None or dict(simtype=1|2, radius, centre, height, v) -- the added droplet / jet.
    phys: None or dict with any of rho0, Cs, alpha, gravity (the reference's constants otherwise)."""
    c = Case()
    for k, v in (phys or {}).items():
        setattr(c, k, float(v))
    if drop:
        c.simtype = int(drop["simtype"])
        c.drop_radius = float(drop["radius"])
        c.drop_height = float(drop.get("height", 0.0))
        for d in range(3):
            c.drop_centre[d] = float(drop["centre"][d]) if d < len(drop["centre"]) else 0.0
            c.drop_v[d] = float(drop["v"][d]) if d < len(drop["v"]) else 0.0
    c.variant = variant
    for d in range(3):
        c.npart[d] = int(npart[d]) if d < len(npart) else 1
        c.box[d] = float(box[d]) if d < len(box) else 0.0
        c.start[d] = float(start[d]) if d < len(start) else 0.0
    c.pstep = pstep
    c.dt0 = dt0
    return c


def derive(case):
    k = Consts()
    rc = lib().orc_derive(C.byref(case), C.byref(k))
    if rc != 0:
        raise ValueError("orc_derive failed: %d" % rc)
    return k


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One run of a reference program: `particles`, `particlesh` and the file-scope scalars."""

    FIELDS = ("x", "v", "V", "f", "rho", "p", "Rrho", "m")

    def __init__(self, case, search=SEARCH_CELLS):
        self.case = case
        self._h = lib().orc_create(C.byref(case))
        if not self._h:
            raise ValueError("orc_create failed")
        self.k = Consts()
        lib().orc_get_consts(self._h, C.byref(self.k))
        self.dim = self.k.dim
        self.npts = int(self.k.npts)
        self.bound_parts = int(self.k.bound_parts)
        lib().orc_set_search(self._h, search)

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_state(self, which):
        n, d = self.npts, self.dim
        out = {k: np.empty((n, d), np.float64) for k in ("x", "v", "V", "f")}
        out.update({k: np.empty(n, np.float32) for k in ("rho", "p", "Rrho", "m")})
        lib().orc_get_state(self._h, which, *[_ptr(out[k]) for k in self.FIELDS])
        return out

    def set_state(self, which, **fields):
        args = []
        for k in self.FIELDS:
            a = fields.get(k)
            if a is not None:
                dt = np.float64 if k in ("x", "v", "V", "f") else np.float32
                a = np.ascontiguousarray(a, dtype=dt)
                fields[k] = a
            args.append(_ptr(a))
        lib().orc_set_state(self._h, which, *args)

    def forces(self, which):
        lib().orc_forces(self._h, which)

    def zero_boundary_f(self, which):
        lib().orc_zero_boundary_f(self._h, which)

    def step(self, nsteps=1, max_sweeps=-1):
        lib().orc_step(self._h, nsteps, max_sweeps)

    def density_reinit(self):
        lib().orc_density_reinit(self._h)

    def set_dt(self, dt):
        lib().orc_set_dt(self._h, dt)

    def scalars(self):
        dt, maxmu, maxf, errsum, logbase = (C.c_double() for _ in range(5))
        t = C.c_float()
        lib().orc_get_scalars(self._h, C.byref(dt), C.byref(t), C.byref(maxmu), C.byref(maxf),
                              C.byref(errsum), C.byref(logbase))
        return dict(dt=dt.value, t=t.value, maxmu=maxmu.value, maxf=maxf.value,
                    errsum=errsum.value, logbase=logbase.value)

    def neighbours(self, which, method=SEARCH_CELLS):
        offs = np.zeros(self.npts + 1, np.uint64)
        lib().orc_neighbours(self._h, which, method, _ptr(offs), None)
        cols = np.zeros(int(offs[-1]), np.uint32)
        lib().orc_neighbours(self._h, which, method, _ptr(offs), _ptr(cols))
        return offs, cols

    def w2kernel(self, r):
        return lib().orc_w2kernel(self._h, float(r))

    def w2gradk(self, rij, r):
        rij = np.ascontiguousarray(rij, np.float64)
        out = np.zeros(self.dim, np.float64)
        lib().orc_w2gradk(self._h, _ptr(rij), float(r), _ptr(out))
        return out
