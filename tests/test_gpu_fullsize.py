"""GPU tests at the benchmark's full size (3-D dam break, ~12.7 M particles), where the CPU
oracle is too slow: size-independent properties and known answers.

  * t=0 Forces: with v = 0, p = 0, rho = rho0 every pair term vanishes, so f = g exactly on the
    fluid, Rrho = 0, V = 0, maxmu = 0 (WCXSPH3D.cpp:456-458);
  * first sweep: v+ = 0.5*dt*(g+g), x+ = x + c1*g + c2*g on the fluid, walls unmoved, rho+ =
    rho0, p+ = 0 (WCXSPH3D.cpp:271-277), bit for bit;
  * neighbour sets do not depend on the implementation's knobs (brick size, skin, list
    capacity): per-particle counts equal, the ordered-pair total is n + 2*(unordered pairs);
  * two runs with different knobs (different summation order) agree to 1e-10 after 2 steps.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from sph_amd import capi

import util

pytestmark = pytest.mark.gpu


def dam_break(nx=640, ny=320, nz=48):
    pstep = 0.1
    box = (2 * nx * pstep + 0.3, ny * pstep + 0.3, 2 * nz * pstep)
    return capi.make_case(capi.PROGRAM_WCXSPH3D, (nx, ny, nz), pstep, box, (0.15, 0.15, 0.15))


@pytest.fixture(scope="module")
def big():
    sim = capi.Sim.from_case(dam_break())
    yield sim
    sim.close()


def test_fullsize_t0_and_first_sweep(big):
    sim = big
    nb = sim.bound_parts
    x0 = sim.download(capi.STATE_N, names=("x",))["x"]
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    st = sim.download(capi.STATE_N, names=("f", "V", "Rrho"))
    assert not st["f"][:, :nb].any()
    assert not st["f"][:2, nb:].any() and np.all(st["f"][2, nb:] == -9.81)
    assert not st["Rrho"].any() and not st["V"].any()
    assert sim.scalars()["maxmu"] == 0.0

    sc = sim.sweeps(1)
    st = sim.download(capi.STATE_NP1, names=("x", "v", "rho", "p"))
    dt, g = 0.0002, -9.81
    beta = np.float32(0.25)
    c1 = 0.5 * (dt * dt) * float(np.float32(1) - np.float32(2) * beta)
    c2 = dt * dt * float(beta)
    assert np.all(st["v"][2, nb:] == 0.0 + 0.5 * dt * (g + g))
    assert not st["v"][:2].any() and not st["v"][:, :nb].any()
    assert np.all(st["x"][2, nb:] == x0[2, nb:] + dt * 0.0 + c1 * g + c2 * g)
    assert np.array_equal(st["x"][:2], x0[:2]) and np.array_equal(st["x"][:, :nb], x0[:, :nb])
    assert np.all(st["rho"] == np.float32(1000.0)) and not st["p"].any()
    # sum over the fluid of |x+ - x|^2; each term is d^2 up to the rounding of (x + d) - x
    d = (c1 * g + c2 * g)
    n_fluid = sim.npts - nb
    assert abs(sc["errsum"] - n_fluid * d * d) <= 1e-6 * n_fluid * d * d


def test_fullsize_neighbour_counts_do_not_depend_on_knobs(big):
    offs = np.zeros(big.npts + 1, np.uint64)
    big._chk(capi.lib().sph_neighbours(big._h, capi.STATE_N, offs.ctypes.data, None), "sph_neighbours")
    total = int(offs[-1])
    assert (total - big.npts) % 2 == 0  # every unordered pair appears in two rows
    counts = np.diff(offs.astype(np.int64))
    assert counts.min() >= 1
    assert 27 in counts[big.bound_parts:]  # interior lattice stencil incl. the particle itself

    os.environ["SPH_BRICK_LOG2"] = "1"
    try:
        other = capi.Sim.from_case(dam_break(), skin_frac=0.03, max_neighbours=48)
    finally:
        del os.environ["SPH_BRICK_LOG2"]
    offs2 = np.zeros(other.npts + 1, np.uint64)
    other._chk(capi.lib().sph_neighbours(other._h, capi.STATE_N, offs2.ctypes.data, None), "sph_neighbours")
    assert np.array_equal(offs, offs2)

    # two steps with different cell order, skin and list capacity: same physics to rounding
    for s in (big, other):
        s.forces(capi.STATE_N)
        s.zero_boundary_forces(capi.STATE_N)
    # fresh state for `big` (the previous test left it inside a step)
    p, st = capi.case_init(dam_break())
    big.upload(capi.STATE_N, st)
    big.upload(capi.STATE_NP1, st)
    big.set_dt(0.0002)
    big.forces(capi.STATE_N)
    big.zero_boundary_forces(capi.STATE_N)
    a = big.step(2)
    b = other.step(2)
    assert abs(a["dt"] - b["dt"]) <= 1e-12 * a["dt"]
    assert a["hit_pairs"] == b["hit_pairs"]
    ga = big.download(capi.STATE_NP1, names=("x", "v", "f", "rho"))
    gb = other.download(capi.STATE_NP1, names=("x", "v", "f", "rho"))
    for k in ("x", "v", "f"):
        assert np.isfinite(ga[k]).all()
        scale = np.max(np.abs(ga[k]))
        assert np.max(np.abs(ga[k] - gb[k])) <= 1e-10 * scale, k
    assert np.max(np.abs(ga["rho"].astype(np.float64) - gb["rho"])) <= 2e-4  # a few float ulp of 1000
    other.close()


def test_c4_on_one_gpu_t0_first_sweep_and_neighbour_counts():
    """The >=100 M-particle configuration (fluid 2048x1024x48 + walls = 121.5 M particles) fits one
    GPU: the same known answers as at C3 -- f = g at t = 0, the first sweep's closed form bit for
    bit, every particle its own neighbour and the 27-point lattice stencil present."""
    sim = capi.Sim.from_case(dam_break(2048, 1024, 48))
    try:
        assert sim.npts > 100_000_000
        nb = sim.bound_parts
        x0 = sim.download(capi.STATE_N, names=("x",))["x"]
        sim.forces(capi.STATE_N)
        sim.zero_boundary_forces(capi.STATE_N)
        st = sim.download(capi.STATE_N, names=("f", "V", "Rrho"))
        assert not st["f"][:, :nb].any()
        assert not st["f"][:2, nb:].any() and np.all(st["f"][2, nb:] == -9.81)
        assert not st["Rrho"].any() and not st["V"].any()
        assert sim.scalars()["maxmu"] == 0.0
        del st

        offs = np.zeros(sim.npts + 1, np.uint64)
        sim._chk(capi.lib().sph_neighbours(sim._h, capi.STATE_N, offs.ctypes.data, None), "sph_neighbours")
        counts = np.diff(offs.astype(np.int64))
        assert (int(offs[-1]) - sim.npts) % 2 == 0
        assert counts.min() >= 1 and 27 in counts[nb:nb + 4_000_000]
        del offs, counts

        sc = sim.sweeps(1)
        st = sim.download(capi.STATE_NP1, names=("x", "v", "rho", "p"))
        dt, g = 0.0002, -9.81
        beta = np.float32(0.25)
        c1 = 0.5 * (dt * dt) * float(np.float32(1) - np.float32(2) * beta)
        c2 = dt * dt * float(beta)
        assert np.all(st["v"][2, nb:] == 0.0 + 0.5 * dt * (g + g))
        assert not st["v"][:2].any() and not st["v"][:, :nb].any()
        assert np.all(st["x"][2, nb:] == x0[2, nb:] + dt * 0.0 + c1 * g + c2 * g)
        assert np.array_equal(st["x"][:2], x0[:2]) and np.array_equal(st["x"][:, :nb], x0[:, :nb])
        assert np.all(st["rho"] == np.float32(1000.0)) and not st["p"].any()
        d = (c1 * g + c2 * g)
        n_fluid = sim.npts - nb
        assert abs(sc["errsum"] - n_fluid * d * d) <= 1e-6 * n_fluid * d * d
    finally:
        sim.close()


# ---- 2-D WCSPH at ~1 M particles (single mass, no XSPH, radius^2 = 4 H^2 + 0.001 H) -------------

def dam_break_2d():
    pstep = 0.003
    return capi.make_case(capi.PROGRAM_WCSPH, (833, 1267), pstep, (15.0, 8.0), (0.31, 0.31))


def test_wcsph_million_particles_known_answers_and_knob_independence():
    case = dam_break_2d()
    p, st0 = capi.case_init(case)
    dt = 0.3 * p.H / p.Cs
    case.dt0 = dt
    sim = capi.Sim.from_case(case)
    assert sim.npts > 1_000_000
    nb = sim.bound_parts
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    st = sim.download(capi.STATE_N, names=("f", "Rrho"))
    assert not st["f"][:, :nb].any() and not st["f"][0].any() and np.all(st["f"][1, nb:] == -9.81)
    assert not st["Rrho"].any() and sim.scalars()["maxmu"] == 0.0

    offs = np.zeros(sim.npts + 1, np.uint64)
    sim._chk(capi.lib().sph_neighbours(sim._h, capi.STATE_N, offs.ctypes.data, None), "sph_neighbours")
    counts = np.diff(offs.astype(np.int64))
    assert (int(offs[-1]) - sim.npts) % 2 == 0 and counts.min() >= 1
    # interior stencil: radius^2 = 4 H^2 + 0.001 H (WCSPH.cpp:96) does not scale with pstep; at
    # pstep 0.003 it is (2.04 pstep)^2, which takes in (+-2,0) and (0,+-2): 13 points, not the 9
    # of the program as shipped
    assert np.bincount(counts[nb:]).argmax() == 13

    sc = sim.sweeps(1)
    g = -9.81
    beta = np.float32(0.25)
    c1 = 0.5 * (dt * dt) * float(np.float32(1) - np.float32(2) * beta)
    c2 = dt * dt * float(beta)
    s1 = sim.download(capi.STATE_NP1, names=("x", "v", "rho", "p"))
    assert np.all(s1["v"][1, nb:] == 0.0 + 0.5 * dt * (g + g)) and not s1["v"][0].any()
    assert np.all(s1["x"][1, nb:] == st0["x"][1, nb:] + dt * 0.0 + c1 * g + c2 * g)
    assert np.array_equal(s1["x"][0], st0["x"][0])
    assert np.all(s1["rho"] == np.float32(1000.0)) and not s1["p"].any()

    os.environ["SPH_BRICK_LOG2"] = "2"
    try:
        other = capi.Sim.from_case(case, skin_frac=0.04, max_neighbours=16)
    finally:
        del os.environ["SPH_BRICK_LOG2"]
    sim.upload(capi.STATE_NP1, st0)
    for s in (sim, other):
        s.set_dt(dt)
        s.forces(capi.STATE_N)
        s.zero_boundary_forces(capi.STATE_N)
    a, b = sim.step(3), other.step(3)
    assert a["hit_pairs"] == b["hit_pairs"] and abs(a["dt"] - b["dt"]) <= 1e-12 * a["dt"]
    ga = sim.download(capi.STATE_NP1, names=("x", "v", "f"))
    gb = other.download(capi.STATE_NP1, names=("x", "v", "f"))
    for k in ("x", "v", "f"):
        assert np.isfinite(ga[k]).all()
        assert np.max(np.abs(ga[k] - gb[k])) <= 1e-10 * np.max(np.abs(ga[k])), k
    sim.close()
    other.close()


# ---- the oracle at the configurations' real sizes (one step each: the oracle does ~65 k
# ---- particle-steps/s in 3-D, ~150 k in 2-D) ---------------------------------------------------

def _one_step_against_the_oracle(spec, what, min_particles):
    oc, pc = util.both_cases(spec)
    orc, sim = O.Oracle(oc), capi.Sim.from_case(pc)
    assert sim.npts == orc.npts > min_particles
    # neighbour sets of the initial state: bit-exact (the oracle's cell back-end)
    go, gc = sim.neighbours(capi.STATE_N)
    oo, occ = orc.neighbours(O.STATE_N)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ), what + ": neighbour sets"
    for s_, which in ((orc, O.STATE_N), (sim, capi.STATE_N)):
        s_.forces(which)
    orc.zero_boundary_f(O.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    orc.step(1)
    sc = sim.step(1)
    g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
    for k in ("x", "v", "V", "f"):
        err = util.rel_err(g[k].T, o[k])
        assert err <= 1e-10, "%s: field %s off by %.3e" % (what, k, err)
    for k in ("rho", "p"):
        assert util.float_ulp_diff(g[k], o[k]).max() <= 1, "%s: %s" % (what, k)
    osc = orc.scalars()
    for k in ("dt", "maxmu", "maxf"):
        assert abs(sc[k] - osc[k]) <= 1e-12 * max(abs(osc[k]), 1e-300), "%s: %s" % (what, k)
    # the sets after the step (particles have moved; the walls' neighbours are in)
    go, gc = sim.neighbours(capi.STATE_NP1)
    oo, occ = orc.neighbours(O.STATE_NP1)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ), what + ": neighbour sets after the step"
    # DensityReinit on the stepped state (WCXSPH3D.cpp:208-249): at this size the whole walk is one
    # launch; rho within one float ulp of the oracle's sequential walk, nothing else touched
    orc.density_reinit()
    sim.density_reinit()
    g2, o2 = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
    u = util.float_ulp_diff(g2["rho"], o2["rho"])
    print("%s: DensityReinit rho max ulp %d, differing %d of %d" % (what, u.max(), int((u > 0).sum()), len(u)))
    assert u.max() <= 1, what + ": DensityReinit"
    assert np.array_equal(g2["x"], g["x"]) and np.array_equal(g2["p"].view(np.int32), g["p"].view(np.int32))
    sim.close()
    orc.close()


def test_c2_wcsph_full_size_one_step_against_the_oracle():
    pstep = 0.003
    p, _ = capi.case_init(dam_break_2d())
    spec = dict(program=0, npart=(833, 1267), pstep=pstep, box=(15.0, 8.0), start=(0.31, 0.31), dt0=0.3 * p.H / p.Cs)
    _one_step_against_the_oracle(spec, "C2 (2-D WCSPH, 1.08 M)", 1_000_000)


def test_3d_million_particle_block_one_step_against_the_oracle():
    spec = dict(program=2, npart=(160, 96, 48), pstep=0.1, box=(32.3, 9.9, 9.6), start=(0.15, 0.15, 0.15))
    _one_step_against_the_oracle(spec, "3-D WCXSPH3D block, ~1 M", 900_000)
