"""ctypes binding of libsph_amd.so (include/sph_amd.h) and a thin host-side mirror of the
reference's driver objects: `Sim` owns one context and speaks in the reference's terms
(`particles` = state n, `particlesh` = state n+1, Forces, Newmark_Beta, DensityReinit).

There is no CPU fallback: if the HIP library is missing or no GPU is present, loading or
creating a context raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPH_AMD_LIB: another build of the same library (development experiments)
LIB_PATH = os.environ.get("SPH_AMD_LIB") or os.path.join(_HERE, "lib", "libsph_amd.so")

OK, EINVAL, EDEVICE, ENOMEM, ESTATE, ECOMM = range(6)
VARIANT_WCSPH, VARIANT_WCXSPH = 0, 1
STATE_N, STATE_NP1 = 0, 1
REINIT_NONE, REINIT_ONE_WORKGROUP, REINIT_LEVELS, REINIT_ONE_LAUNCH, REINIT_ONE_LAUNCH_ROUNDS = range(5)
PROGRAM_WCSPH, PROGRAM_WCXSPH, PROGRAM_WCXSPH3D = 0, 1, 2

# every symbol include/sph_amd.h declares
SYMBOLS = (
    "sph_create", "sph_destroy", "sph_last_error", "sph_upload", "sph_download", "sph_forces",
    "sph_zero_boundary_forces", "sph_step", "sph_step_noalloc", "sph_sweeps", "sph_get_scalars", "sph_set_dt",
    "sph_density_reinit", "sph_neighbours", "sph_profile_enable", "sph_profile_read", "sph_profile_read_kinds",
    "sph_profile_read_comm", "sph_reinit_info",
    "sph_case_default", "sph_case_params", "sph_case_init",
    "sph_set_transport", "sph_upload_global_ids", "sph_rccl_unique_id", "sph_use_rccl",
    "sph_case_init_range", "sph_case_histogram_x", "sph_local_count", "sph_download_global_ids",
    "sph_set_slab_interval", "sph_halo_counts", "sph_rccl_selftest",
)


class Params(C.Structure):
    _fields_ = [("dim", C.c_int32), ("variant", C.c_int32), ("npts", C.c_uint64),
                ("bound_parts", C.c_uint64), ("H", C.c_double), ("HSQ", C.c_double),
                ("correc", C.c_double), ("search_radius_sq", C.c_double), ("B", C.c_double),
                ("gam", C.c_double), ("rho0", C.c_double), ("Cs", C.c_double),
                ("alpha", C.c_double), ("eps", C.c_double), ("g", C.c_double),
                ("g_axis", C.c_int32), ("subits", C.c_int32), ("beta", C.c_double),
                ("mass", C.c_double), ("dt0", C.c_double), ("skin_frac", C.c_double),
                ("max_neighbours", C.c_int32), ("device", C.c_int32),
                ("rank", C.c_int32), ("nranks", C.c_int32), ("npts_global", C.c_uint64),
                ("capacity", C.c_uint64), ("slab_lo", C.c_double), ("slab_hi", C.c_double),
                ("migrate", C.c_int32), ("reserved_", C.c_int32), ("inner_skin_frac", C.c_double)]


class Fields(C.Structure):
    _fields_ = [("x", C.c_void_p * 3), ("v", C.c_void_p * 3), ("V", C.c_void_p * 3),
                ("f", C.c_void_p * 3), ("rho", C.c_void_p), ("p", C.c_void_p),
                ("Rrho", C.c_void_p), ("m", C.c_void_p)]


class Scalars(C.Structure):
    _fields_ = [("dt", C.c_double), ("t", C.c_float), ("maxmu", C.c_double), ("maxf", C.c_double),
                ("errsum", C.c_double), ("logbase", C.c_double), ("list_rebuilds", C.c_uint64),
                ("step_redos", C.c_uint64), ("hit_pairs", C.c_uint64), ("host_syncs", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Case(C.Structure):
    _fields_ = [("program", C.c_int32), ("npart", C.c_int32 * 3), ("pstep", C.c_double),
                ("box", C.c_double * 3), ("start", C.c_double * 3), ("dt0", C.c_double),
                ("rho0", C.c_double), ("Cs", C.c_double), ("alpha", C.c_double), ("gravity", C.c_double),
                ("simtype", C.c_int32), ("reserved_", C.c_int32), ("drop_radius", C.c_double),
                ("drop_centre", C.c_double * 3), ("drop_height", C.c_double), ("drop_v", C.c_double * 3)]


SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                         C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_void_p)
ALLREDUCE2_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.c_int32,
                           C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("host_buffers", C.c_int32), ("struct_size", C.c_uint32),
                ("sendrecv", SENDRECV_FN), ("allreduce", ALLREDUCE_FN), ("allreduce_max_sum", ALLREDUCE2_FN)]


def make_transport(host_buffers, sendrecv, allreduce, allreduce_max_sum=None):
    """A zeroed sph_transport with its size and the callbacks filled in."""
    t = Transport()
    t.host_buffers = host_buffers
    t.struct_size = C.sizeof(Transport)
    t.sendrecv = sendrecv
    t.allreduce = allreduce
    if allreduce_max_sum is not None:
        t.allreduce_max_sum = allreduce_max_sum
    return t


class SphError(RuntimeError):
    """A call of the library returned a code other than SPH_OK; `code` holds it."""

    def __init__(self, msg, code=None):
        RuntimeError.__init__(self, msg)
        self.code = code


_lib = None


def lib():
    """Load libsph_amd.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SphError("libsph_amd.so is not built (run `make -C sph_amd/csrc` or "
                       "__graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.sph_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.sph_destroy.argtypes = [vp]
    L.sph_destroy.restype = None
    L.sph_last_error.argtypes = [vp]
    L.sph_last_error.restype = C.c_char_p
    L.sph_upload.argtypes = [vp, C.c_int32, C.POINTER(Fields)]
    L.sph_download.argtypes = [vp, C.c_int32, C.POINTER(Fields)]
    L.sph_forces.argtypes = [vp, C.c_int32]
    L.sph_zero_boundary_forces.argtypes = [vp, C.c_int32]
    L.sph_step.argtypes = [vp, C.c_int32, C.POINTER(Scalars)]
    L.sph_sweeps.argtypes = [vp, C.c_int32, C.POINTER(Scalars)]
    L.sph_step_noalloc.argtypes = [vp, C.c_int32, C.POINTER(Scalars)]
    L.sph_get_scalars.argtypes = [vp, C.POINTER(Scalars)]
    L.sph_set_dt.argtypes = [vp, C.c_double]
    L.sph_density_reinit.argtypes = [vp]
    L.sph_neighbours.argtypes = [vp, C.c_int32, vp, vp]
    L.sph_profile_enable.argtypes = [vp, C.c_int32]
    L.sph_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.sph_profile_read_kinds.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.sph_reinit_info.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.sph_profile_read_comm.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.sph_case_default.argtypes = [C.c_int32, C.POINTER(Case)]
    L.sph_case_params.argtypes = [C.POINTER(Case), C.POINTER(Params), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]
    L.sph_case_init.argtypes = [C.POINTER(Case), C.POINTER(Fields)]
    L.sph_set_transport.argtypes = [vp, C.POINTER(Transport)]
    L.sph_upload_global_ids.argtypes = [vp, vp]
    L.sph_local_count.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sph_download_global_ids.argtypes = [vp, vp]
    L.sph_set_slab_interval.argtypes = [vp, C.c_double, C.c_double]
    L.sph_halo_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sph_rccl_unique_id.argtypes = [vp]
    L.sph_rccl_selftest.argtypes = [C.c_int32, C.c_char_p, C.c_int32]
    L.sph_use_rccl.argtypes = [vp, vp]
    L.sph_case_init_range.argtypes = [C.POINTER(Case), C.c_double, C.c_double, C.POINTER(Fields), vp,
                                      C.POINTER(C.c_uint64)]
    L.sph_case_histogram_x.argtypes = [C.POINTER(Case), C.c_int32, vp, vp]
    _lib = L
    return L


# ---------------------------------------------------------------------------------------------
# cases


def rccl_unique_id():
    buf = (C.c_char * 128)()
    if lib().sph_rccl_unique_id(buf) != OK:
        raise SphError("sph_rccl_unique_id failed (is librccl.so.1 on the loader path?)")
    return bytes(buf.raw)


def rccl_selftest(device=-1):
    """Runs the library's RCCL calls on one GPU (a one-rank communicator); returns its message."""
    buf = C.create_string_buffer(256)
    rc = lib().sph_rccl_selftest(device, buf, 256)
    return rc, buf.value.decode()


def default_case(program):
    c = Case()
    if lib().sph_case_default(program, C.byref(c)) != OK:
        raise SphError("unknown program %r" % (program,))
    return c


SIMTYPE_BOX, SIMTYPE_DROPLET, SIMTYPE_JET = 0, 1, 2


def make_case(program, npart, pstep, box, start, dt0=0.0002, drop=None, phys=None):
    """drop: None or dict(simtype=1|2, radius, centre, height, v) -- a droplet / jet next to the
    reference's block (no counterpart in the reference).
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
    c.program = program
    for d in range(3):
        c.npart[d] = int(npart[d]) if d < len(npart) else 1
        c.box[d] = float(box[d]) if d < len(box) else 0.0
        c.start[d] = float(start[d]) if d < len(start) else 0.0
    c.pstep = float(pstep)
    c.dt0 = float(dt0)
    return c


def case_params(case):
    p = Params()
    sm, bm = C.c_double(), C.c_double()
    if lib().sph_case_params(C.byref(case), C.byref(p), C.byref(sm), C.byref(bm)) != OK:
        raise SphError("invalid case")
    return p, sm.value, bm.value


_VEC = ("x", "v", "V", "f")
_SCA = ("rho", "p", "Rrho", "m")


def alloc_state(npts, dim):
    """Host arrays of one State, component-major (x[d] is contiguous)."""
    s = {k: np.zeros((dim, npts), np.float64) for k in _VEC}
    s.update({k: np.zeros(npts, np.float32) for k in _SCA})
    return s


def _fields(state, dim, names=None):
    f = Fields()
    for k in _VEC:
        a = state.get(k) if (names is None or k in names) else None
        if a is None:
            continue
        assert a.dtype == np.float64 and a.flags.c_contiguous and a.shape[0] == dim
        arr = getattr(f, k)
        for d in range(dim):
            arr[d] = a[d].ctypes.data
    for k in _SCA:
        a = state.get(k) if (names is None or k in names) else None
        if a is None:
            continue
        assert a.dtype == np.float32 and a.flags.c_contiguous
        setattr(f, k, a.ctypes.data)
    return f


def case_init(case):
    """InitSPH: returns (params, state dict in original order)."""
    p, _, _ = case_params(case)
    st = alloc_state(int(p.npts), p.dim)
    f = _fields(st, p.dim)
    if lib().sph_case_init(C.byref(case), C.byref(f)) != OK:
        raise SphError("sph_case_init failed")
    return p, st


def case_init_range(case, xlo, xhi):
    """The particles of InitSPH with xlo <= x < xhi: (state dict, global indices)."""
    p, _, _ = case_params(case)
    cnt = C.c_uint64()
    if lib().sph_case_init_range(C.byref(case), xlo, xhi, None, None, C.byref(cnt)) != OK:
        raise SphError("sph_case_init_range failed")
    n = int(cnt.value)
    st = alloc_state(n, p.dim)
    gids = np.zeros(n, np.uint32)
    f = _fields(st, p.dim)
    if lib().sph_case_init_range(C.byref(case), xlo, xhi, C.byref(f), gids.ctypes.data, C.byref(cnt)) != OK:
        raise SphError("sph_case_init_range failed")
    return st, gids


def case_histogram_x(case, nbins):
    fl, wl = np.zeros(nbins, np.uint64), np.zeros(nbins, np.uint64)
    if lib().sph_case_histogram_x(C.byref(case), nbins, fl.ctypes.data, wl.ctypes.data) != OK:
        raise SphError("sph_case_histogram_x failed")
    return fl, wl


# ---------------------------------------------------------------------------------------------


class Sim:
    """One run on one GPU: both State vectors and the file-scope scalars of the reference."""

    def __init__(self, params):
        self.params = params
        self.dim = params.dim
        self.npts = int(params.npts)
        self.bound_parts = int(params.bound_parts)
        h = C.c_void_p()
        rc = lib().sph_create(C.byref(params), C.byref(h))
        self._h = h
        if rc != OK:
            msg = lib().sph_last_error(h).decode() if h else "sph_create failed"
            if h:
                lib().sph_destroy(h)
            self._h = None
            raise SphError("sph_create: %s (code %d)" % (msg, rc))

    @classmethod
    def from_case(cls, case, **overrides):
        """main() up to the kd-tree construction: InitSPH, copy into particlesh."""
        p, st = case_init(case)
        for k, v in overrides.items():
            setattr(p, k, v)
        sim = cls(p)
        sim.upload(STATE_N, st)
        sim.upload(STATE_NP1, st)
        return sim

    def _chk(self, rc, what):
        if rc != OK:
            raise SphError("%s: %s (code %d)" % (what, lib().sph_last_error(self._h).decode(), rc), rc)

    def close(self):
        if getattr(self, "_h", None):
            lib().sph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, which, state, names=None):
        st = {}
        for k, a in state.items():
            if a is None:
                continue
            st[k] = np.ascontiguousarray(a, np.float64 if k in _VEC else np.float32)
        f = _fields(st, self.dim, names)
        self._chk(lib().sph_upload(self._h, which, C.byref(f)), "sph_upload")

    def download(self, which, names=None):
        st = alloc_state(self.npts, self.dim)
        f = _fields(st, self.dim, names)
        self._chk(lib().sph_download(self._h, which, C.byref(f)), "sph_download")
        if names is not None:
            st = {k: v for k, v in st.items() if k in names}
        return st

    def set_transport(self, transport):
        """Install a `Transport`; the object (and its callbacks) is kept alive by this Sim."""
        self._transport = transport
        self._chk(lib().sph_set_transport(self._h, C.byref(transport)), "sph_set_transport")

    def use_rccl(self, unique_id):
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._chk(lib().sph_use_rccl(self._h, buf), "sph_use_rccl")

    def upload_global_ids(self, gids):
        g = np.ascontiguousarray(gids, np.uint32)
        assert g.shape == (self.npts,)
        self._chk(lib().sph_upload_global_ids(self._h, g.ctypes.data), "sph_upload_global_ids")

    def local_count(self):
        n, mi, mo = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(lib().sph_local_count(self._h, C.byref(n), C.byref(mi), C.byref(mo)), "sph_local_count")
        return int(n.value), int(mi.value), int(mo.value)

    def halo_counts(self):
        """Particles sent to / received from the (left, right) neighbour at every sweep."""
        s, r = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        self._chk(lib().sph_halo_counts(self._h, s, r), "sph_halo_counts")
        return (int(s[0]), int(s[1])), (int(r[0]), int(r[1]))

    def set_slab_interval(self, lo, hi):
        self._chk(lib().sph_set_slab_interval(self._h, lo, hi), "sph_set_slab_interval")

    def download_global_ids(self):
        g = np.zeros(self.npts, np.uint32)
        self._chk(lib().sph_download_global_ids(self._h, g.ctypes.data), "sph_download_global_ids")
        return g

    def forces(self, which):
        self._chk(lib().sph_forces(self._h, which), "sph_forces")

    def zero_boundary_forces(self, which):
        self._chk(lib().sph_zero_boundary_forces(self._h, which), "sph_zero_boundary_forces")

    def step(self, nsteps=1):
        s = Scalars()
        self._chk(lib().sph_step(self._h, nsteps, C.byref(s)), "sph_step")
        return s.as_dict()

    def step_noalloc(self, nsteps=1):
        """sph_step through the entry point that allocates nothing."""
        s = Scalars()
        self._chk(lib().sph_step_noalloc(self._h, nsteps, C.byref(s)), "sph_step_noalloc")
        return s.as_dict()

    def sweeps(self, nsweeps):
        s = Scalars()
        self._chk(lib().sph_sweeps(self._h, nsweeps, C.byref(s)), "sph_sweeps")
        return s.as_dict()

    def scalars(self):
        s = Scalars()
        self._chk(lib().sph_get_scalars(self._h, C.byref(s)), "sph_get_scalars")
        return s.as_dict()

    def set_dt(self, dt):
        self._chk(lib().sph_set_dt(self._h, dt), "sph_set_dt")

    def density_reinit(self):
        self._chk(lib().sph_density_reinit(self._h), "sph_density_reinit")

    def reinit_info(self):
        """How the last density_reinit() walked: dict(walk=one of REINIT_*, launches, rounds, took_turns,
        standing_lists: no list rebuild was needed)."""
        out = (C.c_uint64 * 4)()
        self._chk(lib().sph_reinit_info(self._h, out), "sph_reinit_info")
        return dict(walk=int(out[0]), launches=int(out[1]), rounds=int(out[2]), took_turns=bool(out[3] & 1),
                    standing_lists=bool(out[3] & 2))

    def neighbours(self, which):
        offs = np.zeros(self.npts + 1, np.uint64)
        self._chk(lib().sph_neighbours(self._h, which, offs.ctypes.data, None), "sph_neighbours")
        cols = np.zeros(max(int(offs[-1]), 1), np.uint32)
        self._chk(lib().sph_neighbours(self._h, which, offs.ctypes.data, cols.ctypes.data),
                  "sph_neighbours")
        return offs, cols[:int(offs[-1])]

    def profile_enable(self, on=True):
        self._chk(lib().sph_profile_enable(self._h, 1 if on else 0), "sph_profile_enable")

    def profile_read(self):
        """Pair-kernel and rebuild durations since the last read; `by_kind`: pair launches that
        walked the candidate lists / pair launches that walked the step lists / draws of the
        step lists."""
        kms, kn = (C.c_double * 3)(), (C.c_uint64 * 3)()
        self._chk(lib().sph_profile_read_kinds(self._h, kms, kn), "sph_profile_read_kinds")
        cms, cn = (C.c_double * 4)(), (C.c_uint64 * 4)()
        self._chk(lib().sph_profile_read_comm(self._h, cms, cn), "sph_profile_read_comm")
        out = self._profile_read()
        out["pair_ms_by_kind"] = [float(v) for v in kms]
        out["pair_launches_by_kind"] = [int(v) for v in kn]
        # slab runs: halo exchange of a sweep, the step's all-reduce, halo set-up and migration at a rebuild
        out["comm_ms"] = [float(v) for v in cms]
        out["comm_counts"] = [int(v) for v in cn]
        return out

    def _profile_read(self):
        pm, rm = C.c_double(), C.c_double()
        pn, rn = C.c_uint64(), C.c_uint64()
        self._chk(lib().sph_profile_read(self._h, C.byref(pm), C.byref(pn), C.byref(rm), C.byref(rn)),
                  "sph_profile_read")
        return dict(pair_ms=pm.value, pair_launches=pn.value, rebuild_ms=rm.value, rebuilds=rn.value)
