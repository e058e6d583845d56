"""Shared helpers of the parity tests: case tables and comparisons."""
import numpy as np

from oracle import oracle as O
from sph_amd import capi

# (oracle variant id, product program id) share the same numbering
PROGRAMS = {"wcsph": 0, "wcxsph": 1, "wcxsph3d": 2}


def small_cases():
    """The three programs as shipped plus modest scaled versions."""
    return {
        "wcsph_default": dict(program=0),
        "wcxsph_default": dict(program=1),
        "wcxsph3d_default": dict(program=2),
        # fluid touching the walls sooner, a few thousand particles
        "wcxsph3d_block": dict(program=2, npart=(12, 10, 8), pstep=0.1, box=(2.0, 1.5, 1.2),
                               start=(0.1, 0.1, 0.1)),
        "wcsph_fine": dict(program=0, npart=(60, 90), pstep=0.04, box=(6.0, 4.0), start=(0.1, 0.1)),
        "wcxsph_fine": dict(program=1, npart=(80, 60), pstep=0.05, box=(6.0, 4.0), start=(0.06, 0.06)),
    }


def both_cases(spec):
    """Returns (oracle case, product case) for one spec."""
    prog = spec["program"]
    if "npart" not in spec:
        return O.default_case(prog), capi.default_case(prog)
    args = (spec["npart"], spec["pstep"], spec["box"], spec["start"], spec.get("dt0", 0.0002))
    drop = spec.get("drop")
    phys = spec.get("phys")
    return O.make_case(prog, *args, drop=drop, phys=phys), capi.make_case(prog, *args, drop=drop, phys=phys)


def drop_cases():
    """Bodies the reference's InitSPH cannot make: a droplet falling into a shallow pool and a
    jet crossing the box (non-uniform density; with an x-velocity they also cross slab planes)."""
    return {
        "droplet3d": dict(program=2, npart=(16, 10, 3), pstep=0.1, box=(2.0, 1.4, 2.0), start=(0.1, 0.1, 0.1),
                          drop=dict(simtype=1, radius=0.31, centre=(0.95, 0.65, 1.0), v=(0.6, 0.0, -2.0))),
        "jet3d": dict(program=2, npart=(16, 10, 2), pstep=0.1, box=(2.0, 1.4, 2.0), start=(0.1, 0.1, 0.1),
                      drop=dict(simtype=2, radius=0.21, centre=(0.75, 0.65, 1.1), height=1.0, v=(1.5, 0.0, -1.0))),
        "droplet2d": dict(program=1, npart=(60, 8), pstep=0.05, box=(4.0, 3.0), start=(0.06, 0.06),
                          drop=dict(simtype=1, radius=0.26, centre=(1.9, 1.5), v=(0.8, -2.0))),
    }


def to_particle_major(st):
    """product state (component-major) -> oracle layout (particle-major)."""
    out = {}
    for k, a in st.items():
        out[k] = np.ascontiguousarray(a.T) if a.ndim == 2 else a
    return out


def to_component_major(st):
    out = {}
    for k, a in st.items():
        out[k] = np.ascontiguousarray(a.T) if a.ndim == 2 else a
    return out


def float_ulp_diff(a, b):
    """Distance in float32 units in the last place."""
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    return np.abs(ia - ib)


def rel_err(a, b, scale=None):
    """max |a-b| / scale with scale = max|b| by default (field-wise relative error)."""
    s = np.max(np.abs(b)) if scale is None else scale
    if s == 0:
        return float(np.max(np.abs(a - b)))
    return float(np.max(np.abs(a - b)) / s)


def jitter(x, pstep, seed=12345, amp=0.01):
    rng = np.random.Generator(np.random.MT19937(seed))
    return x + rng.uniform(-amp * pstep, amp * pstep, size=x.shape)
