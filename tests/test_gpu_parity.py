"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bars (stated once, used throughout):
  * neighbour sets: bit-exact (CSR offsets and columns equal);
  * double fields x, v, V, f: max|a-b| <= 1e-10 * max|b| per field, for one Forces call or one
    sweep from identical inputs; the same bound is asserted after 10 full steps on the small
    cases (measured drift is ~1e-15) and the measured drift is printed;
  * float fields rho, p: equal, or 1 float ulp apart where the double value lands on a
    rounding tie; Rrho (a float store of a cancelling sum): |a-b| <= 1e-6 * max|b|;
  * scalars dt, maxmu, maxf: relative 1e-12; errsum/logbase (order-dependent sums in the
    reference, diagnostics only): relative 1e-9.
The oracle itself is unpinned by the reference (it ships no fixtures); see oracle/sph_oracle.h.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from sph_amd import capi

import util

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-10
CASES = util.small_cases()


def start(spec, **overrides):
    """main() up to the time loop: InitSPH, copy, Forces at t=0, wall f zeroed -- on both sides."""
    oc, pc = util.both_cases(spec)
    orc = O.Oracle(oc)
    sim = capi.Sim.from_case(pc, **overrides)
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    return orc, sim


def assert_state_close(g, o, what, tol=TOL, ulp=1):
    for k in ("x", "v", "V", "f"):
        err = util.rel_err(g[k].T, o[k])
        assert err <= tol, "%s: field %s off by %.3e" % (what, k, err)
    for k in ("rho", "p"):
        u = util.float_ulp_diff(g[k], o[k])
        assert u.max() <= ulp, "%s: %s differs by %d float ulp" % (what, k, u.max())
    s = max(float(np.max(np.abs(o["Rrho"]))), 1e-30)
    assert float(np.max(np.abs(g["Rrho"].astype(np.float64) - o["Rrho"]))) <= 1e-6 * s, what + ": Rrho"


def assert_scalars_close(sc, osc, what):
    for k in ("dt", "maxmu", "maxf"):
        assert abs(sc[k] - osc[k]) <= 1e-12 * max(abs(osc[k]), 1e-300), "%s: %s %r vs %r" % (what, k, sc[k], osc[k])
    assert abs(sc["errsum"] - osc["errsum"]) <= 1e-9 * max(abs(osc["errsum"]), 1e-300), what
    assert abs(sc["logbase"] - osc["logbase"]) <= 1e-9 * max(abs(osc["logbase"]), 1.0), what
    assert np.float32(sc["t"]) == np.float32(osc["t"]) or abs(sc["t"] - osc["t"]) <= 1e-6 * abs(osc["t"]), what


# ---- neighbour sets ----------------------------------------------------------------------------

@pytest.mark.parametrize("name", list(CASES))
def test_neighbour_sets_exact_at_t0(name):
    oc, pc = util.both_cases(CASES[name])
    orc, sim = O.Oracle(oc), capi.Sim.from_case(pc)
    for gw, ow in ((capi.STATE_N, O.STATE_N), (capi.STATE_NP1, O.STATE_NP1)):
        go, gc = sim.neighbours(gw)
        oo, occ = orc.neighbours(ow)
        assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    sim.close()


@pytest.mark.parametrize("name", ["wcxsph_default", "wcxsph3d_default", "wcsph_fine"])
def test_neighbour_sets_exact_on_jittered_positions(name):
    # jitter moves lattice pairs that sit exactly on the search radius to either side of it
    oc, pc = util.both_cases(CASES[name])
    orc, sim = O.Oracle(oc), capi.Sim.from_case(pc)
    for seed in (12345, 7):
        x = util.jitter(orc.get_state(O.STATE_N)["x"], oc.pstep, seed=seed)
        orc.set_state(O.STATE_NP1, x=x)
        sim.upload(capi.STATE_NP1, {"x": np.ascontiguousarray(x.T)})
        go, gc = sim.neighbours(capi.STATE_NP1)
        oo, occ = orc.neighbours(O.STATE_NP1, O.SEARCH_BRUTE)
        assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    sim.close()


@pytest.mark.parametrize("name", ["wcxsph_default", "wcxsph3d_block"])
def test_neighbour_sets_exact_at_every_sweep(name):
    # the sets Forces sees at the start of each of the 8 sweeps of a step, from identical inputs
    orc, sim = start(CASES[name])
    orc.step(3)
    sim.step(3)
    base = orc.get_state(O.STATE_NP1)
    basen = orc.get_state(O.STATE_N)
    for k in range(1, 8):
        o2 = O.Oracle(util.both_cases(CASES[name])[0])
        o2.set_state(O.STATE_N, **basen)
        o2.set_state(O.STATE_NP1, **base)
        o2.set_dt(orc.scalars()["dt"])
        o2.step(1, k)
        st = o2.get_state(O.STATE_NP1)
        sim.upload(capi.STATE_NP1, {"x": np.ascontiguousarray(st["x"].T)})
        go, gc = sim.neighbours(capi.STATE_NP1)
        oo, occ = o2.neighbours(O.STATE_NP1)
        assert np.array_equal(go, oo) and np.array_equal(gc, occ), "sweep %d" % k
    sim.close()


# ---- Forces and single sweeps from identical inputs --------------------------------------------

@pytest.mark.parametrize("name", list(CASES))
def test_forces_at_t0(name):
    orc, sim = start(CASES[name])
    assert_state_close(sim.download(capi.STATE_N), orc.get_state(O.STATE_N), name)
    assert sim.scalars()["maxmu"] == orc.scalars()["maxmu"] == 0.0
    sim.close()


@pytest.mark.parametrize("name", list(CASES))
def test_forces_on_a_moving_state(name):
    # Forces() alone on a state taken from the middle of a run
    orc, sim = start(CASES[name])
    orc.step(5)
    st = orc.get_state(O.STATE_NP1)
    sim.upload(capi.STATE_NP1, util.to_component_major(st))
    orc.forces(O.STATE_NP1)
    sim.forces(capi.STATE_NP1)
    assert_state_close(sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), name)
    assert abs(sim.scalars()["maxmu"] - orc.scalars()["maxmu"]) <= 1e-12 * max(orc.scalars()["maxmu"], 1e-300)
    sim.close()


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("nsweeps", [1, 2, 8])
def test_sweeps_from_identical_inputs(name, nsweeps):
    orc, sim = start(CASES[name])
    orc.step(4)
    # hand the oracle's exact states to the GPU, then run the same number of sweeps on both
    sim.upload(capi.STATE_N, util.to_component_major(orc.get_state(O.STATE_N)))
    sim.upload(capi.STATE_NP1, util.to_component_major(orc.get_state(O.STATE_NP1)))
    sim.set_dt(orc.scalars()["dt"])
    orc.step(1, nsweeps)
    sc = sim.sweeps(nsweeps)
    assert_state_close(sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), "%s/%d" % (name, nsweeps))
    osc = orc.scalars()
    assert abs(sc["maxmu"] - osc["maxmu"]) <= 1e-12 * max(osc["maxmu"], 1e-300)
    assert abs(sc["errsum"] - osc["errsum"]) <= 1e-9 * max(osc["errsum"], 1e-300)
    sim.close()


# ---- whole steps, drift ------------------------------------------------------------------------

@pytest.mark.parametrize("name", list(CASES))
def test_ten_steps_and_drift(name):
    orc, sim = start(CASES[name])
    worst = 0.0
    for step in range(10):
        orc.step(1)
        sc = sim.step(1)
        assert_scalars_close(sc, orc.scalars(), "%s step %d" % (name, step))
        g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
        worst = max(worst, max(util.rel_err(g[k].T, o[k]) for k in ("x", "v", "V", "f")))
    print("\n%s: drift after 10 steps, worst field-relative error %.3e" % (name, worst))
    for gw, ow in ((capi.STATE_N, O.STATE_N), (capi.STATE_NP1, O.STATE_NP1)):
        assert_state_close(sim.download(gw), orc.get_state(ow), name)
    go, gc = sim.neighbours(capi.STATE_NP1)
    oo, occ = orc.neighbours(O.STATE_NP1)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    assert sc["step_redos"] == 0
    sim.close()


@pytest.mark.parametrize("name", ["wcxsph_default", "wcxsph3d_block", "wcsph_default"])
def test_trajectory_drift_over_200_steps(name):
    # 20 frames of main()'s loop (10 steps + DensityReinit): how far the two trajectories drift
    # apart when every sum is taken in a different order.  Bound asserted: 1e-10 of the field's
    # scale after 200 steps (measured: below 3e-14, rho identical); the measured drift is printed.
    orc, sim = start(CASES[name])
    report = []
    for frame in range(20):
        orc.step(10)
        sim.step(10)
        orc.density_reinit()
        sim.density_reinit()
        # rho straight after every re-initialisation: a neighbour missed or met twice shows here
        # hundreds of steps before it shows in x
        u = int(util.float_ulp_diff(sim.download(capi.STATE_NP1, names=("rho",))["rho"], orc.get_state(O.STATE_NP1)["rho"]).max())
        assert u <= 1, "frame %d: rho %d float ulp from the oracle's" % (frame, u)
        if frame in (0, 4, 9, 19):
            g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
            report.append((10 * (frame + 1), util.rel_err(g["x"].T, o["x"]), util.rel_err(g["v"].T, o["v"]), u))
    print("\n%s: steps, rel err x, rel err v, rho ulp: %s" % (name, report))
    g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
    assert util.rel_err(g["x"].T, o["x"]) <= TOL and util.rel_err(g["v"].T, o["v"]) <= TOL
    go, gc = sim.neighbours(capi.STATE_NP1)
    oo, occ = orc.neighbours(O.STATE_NP1)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    sim.close()


@pytest.mark.parametrize("name,variant", [("wcsph", 0), ("wcxsph", 1), ("wcxsph3d", 2)])
def test_against_golden_vectors(name, variant):
    gold = np.load(os.path.join(GOLDEN, "%s_default_3steps.npz" % name))
    sim = capi.Sim.from_case(capi.default_case(variant))
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    sc = sim.step(int(gold["steps"]))
    g = sim.download(capi.STATE_NP1)
    assert_state_close(g, {k: gold[k] for k in ("x", "v", "V", "f", "rho", "p", "Rrho")}, name)
    assert_scalars_close(sc, {k: float(gold[k]) for k in ("dt", "t", "maxmu", "maxf", "errsum", "logbase")}, name)
    go, gc = sim.neighbours(capi.STATE_NP1)
    assert np.array_equal(go, gold["nbr_offsets"]) and np.array_equal(gc, gold["nbr_cols"])
    sim.close()


# ---- knobs must not change results ---------------------------------------------------------------

def test_tight_skin_forces_rebuilds_and_redos_but_same_answer():
    # a skin this thin is outrun inside a step: the step is redone with fresh lists
    name = "wcxsph3d_block"
    orc, sim = start(CASES[name], skin_frac=1e-4)
    orc.step(6)
    sc = sim.step(6)
    assert sc["list_rebuilds"] >= 6
    assert_state_close(sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), name)
    assert_scalars_close(sc, orc.scalars(), name)
    sim.close()


def test_small_list_capacity_grows():
    name = "wcxsph3d_default"
    orc, sim = start(CASES[name], max_neighbours=8)
    orc.step(2)
    sc = sim.step(2)
    assert_state_close(sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), name)
    sim.close()


def test_moving_walls_keep_wall_wall_pairs():
    # wall particles that move void the shortcut that leaves wall-wall pairs out of the lists
    name = "wcxsph_default"
    oc, pc = util.both_cases(CASES[name])
    orc, sim = O.Oracle(oc), capi.Sim.from_case(pc)
    st = orc.get_state(O.STATE_N)
    st["v"][:orc.bound_parts, 0] = 0.3
    for w in (O.STATE_N, O.STATE_NP1):
        orc.set_state(w, v=st["v"])
    for w in (capi.STATE_N, capi.STATE_NP1):
        sim.upload(w, {"v": np.ascontiguousarray(st["v"].T)})
    orc.forces(O.STATE_N); orc.zero_boundary_f(O.STATE_N)
    sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
    orc.step(3)
    sim.step(3)
    assert_state_close(sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), name)
    sim.close()


def test_a_stray_particle_far_away_keeps_the_sets_exact():
    # one fluid particle flung far out blows the bounding box up; the cell table cannot grow with
    # it, so the cells are stretched to many times the list radius.  The single-precision test of
    # the list build widens its margin with the cell edge: the sets must stay those of the
    # brute-force search, and a step must still agree with the oracle.
    spec = CASES["wcxsph3d_block"]
    orc, sim = start(spec)
    st = orc.get_state(O.STATE_N)
    x = st["x"].copy()
    x[-1] = (3.0e7, 0.6, 0.5)
    for w, gw in ((O.STATE_N, capi.STATE_N), (O.STATE_NP1, capi.STATE_NP1)):
        orc.set_state(w, x=x)
        sim.upload(gw, {"x": np.ascontiguousarray(x.T)})
    go, gc = sim.neighbours(capi.STATE_NP1)
    oo, occ = orc.neighbours(O.STATE_NP1, O.SEARCH_BRUTE)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    orc.step(2)
    sim.step(2)
    assert_state_close(sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), "stray particle")
    go, gc = sim.neighbours(capi.STATE_NP1)
    oo, occ = orc.neighbours(O.STATE_NP1, O.SEARCH_BRUTE)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    sim.close()


# ---- API behaviour -------------------------------------------------------------------------------

def test_upload_download_round_trip_and_errors():
    p, st = capi.case_init(capi.default_case(1))
    sim = capi.Sim(p)
    with pytest.raises(capi.SphError):
        sim.step(1)  # nothing uploaded yet
    rng = np.random.default_rng(3)
    for k in ("v", "f", "V"):
        st[k][:] = rng.normal(size=st[k].shape)
    st["p"][:] = rng.normal(size=st["p"].shape).astype(np.float32)
    for w in (capi.STATE_N, capi.STATE_NP1):
        sim.upload(w, st)
        back = sim.download(w)
        for k in st:
            assert np.array_equal(back[k], st[k]), (w, k)
    # after a rebuild (cell order changes) the original order is still what comes back
    sim.neighbours(capi.STATE_NP1)
    back = sim.download(capi.STATE_NP1)
    for k in st:
        assert np.array_equal(back[k], st[k]), k
    sim.close()
    bad = capi.Params.from_buffer_copy(p)
    bad.dim = 4
    with pytest.raises(capi.SphError):
        capi.Sim(bad)


# ---- larger cases --------------------------------------------------------------------------------

def test_medium_2d_wcsph_one_step():
    # config 2 physics (single mass, no XSPH, wall spacing 0.7 r0) at ~50k particles
    spec = dict(program=0, npart=(180, 260), pstep=0.01, box=(4.0, 3.0), start=(0.03, 0.03), dt0=2e-5)
    orc, sim = start(spec)
    orc.step(1)
    sc = sim.step(1)
    assert_state_close(sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), "wcsph 50k")
    assert_scalars_close(sc, orc.scalars(), "wcsph 50k")
    sim.close()


def test_medium_3d_one_step():
    spec = dict(program=2, npart=(40, 30, 24), pstep=0.1, box=(8.3, 3.3, 4.8), start=(0.15, 0.15, 0.15))
    orc, sim = start(spec)
    orc.step(1)
    sc = sim.step(1)
    assert_state_close(sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), "3d 65k")
    assert_scalars_close(sc, orc.scalars(), "3d 65k")
    go = sim.neighbours(capi.STATE_NP1)
    oo = orc.neighbours(O.STATE_NP1)
    assert np.array_equal(go[0], oo[0]) and np.array_equal(go[1], oo[1])
    sim.close()


# ---- DensityReinit -----------------------------------------------------------------------------
# The reference sums each particle's neighbours in the order its search returns them (by
# distance, ties in an order that depends on nanoflann's internals); the oracle and the HIP path
# each use their own fixed order.  rho is a float store of that sum, so the two can land on
# neighbouring floats, and Gauss-Seidel feeds such a difference into later particles' weights.
# Bar: rho within 1 float ulp after one call (measured: identical), and the usual bars after the
# main loop has gone on from there for three frames.

@pytest.mark.parametrize("name", list(CASES))
def test_density_reinit_after_some_steps(name):
    orc, sim = start(CASES[name])
    orc.step(10)
    sim.step(10)
    orc.density_reinit()
    sim.density_reinit()
    g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
    u = util.float_ulp_diff(g["rho"], o["rho"])
    print("%s: DensityReinit rho max ulp %d, differing %d of %d" % (name, u.max(), int((u > 0).sum()), len(u)))
    assert u.max() <= 1
    # nothing but rho of state n+1 changes (WCXSPH3D.cpp:247)
    for k in ("x", "v", "f"):
        assert util.rel_err(g[k].T, o[k]) <= TOL
    assert np.array_equal(g["p"].view(np.int32), sim.download(capi.STATE_N)["p"].view(np.int32))
    gn, on = sim.download(capi.STATE_N), orc.get_state(O.STATE_N)
    assert util.float_ulp_diff(gn["rho"], on["rho"]).max() <= 1


@pytest.mark.parametrize("name", ["wcxsph_default", "wcxsph3d_block", "wcsph_fine"])
def test_frames_with_density_reinit(name):
    # main()'s loop: 10 steps, DensityReinit, next frame (WCXSPH3D.cpp:475-482)
    orc, sim = start(CASES[name])
    for frame in range(3):
        orc.step(10)
        sim.step(10)
        orc.density_reinit()
        sim.density_reinit()
    g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
    for k in ("x", "v"):
        err = util.rel_err(g[k].T, o[k])
        print("%s after 3 frames: %s rel err %.3e" % (name, k, err))
        assert err <= TOL
    assert util.float_ulp_diff(g["rho"], o["rho"]).max() <= 1


@pytest.mark.parametrize("spec", [
    dict(program=1, npart=(170, 110), pstep=0.05, box=(12.0, 7.0), start=(0.06, 0.06)),
    dict(program=2, npart=(24, 20, 16), pstep=0.1, box=(3.0, 2.4, 2.2), start=(0.15, 0.15, 0.15)),
], ids=["2d_20k", "3d_22k"])
def test_density_reinit_beyond_the_one_workgroup_size(spec):
    # cases of up to 12288 particles are walked by one workgroup; larger ones by a grid of workgroups in one launch
    orc, sim = start(spec)
    assert orc.npts > 16384
    orc.step(4)
    sim.step(4)
    orc.density_reinit()
    sim.density_reinit()
    g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
    assert util.float_ulp_diff(g["rho"], o["rho"]).max() <= 1
    sim.close()


@pytest.mark.parametrize("name", ["wcxsph3d_block", "wcxsph_fine", "droplet3d"])
def test_list_build_from_records_and_by_the_ring_kernel_give_the_same_rows(name, monkeypatch, capfd):
    # The list build records keep masks while it walks the cell ranges and writes the rows from the
    # records; a particle whose records do not fit is handed to the kernel that places keepers in
    # rings as it goes.  Three runs: as shipped (room for 40 records), room for one (most particles
    # handed on), the ring kernel alone.  Rows feed the sums in their order, so every field must come
    # out bit for bit, redone steps and list rebuilds included.
    import re
    spec = util.drop_cases()[name] if name in util.drop_cases() else util.small_cases()[name]
    monkeypatch.setenv("SPH_TRACE", "1")
    got = {}
    for trips in ("40", "1", "0"):
        monkeypatch.setenv("SPH_BUILD_TRIPS", trips)
        _, pc = util.both_cases(spec)
        sim = capi.Sim.from_case(pc)
        sim.forces(capi.STATE_N)
        sim.zero_boundary_forces(capi.STATE_N)
        sc = sim.step(25)
        offs, cols = sim.neighbours(capi.STATE_NP1)
        st = sim.download(capi.STATE_NP1)
        sim.density_reinit()
        rho = sim.download(capi.STATE_NP1, names=("rho",))["rho"]
        sim.close()
        handed = [int(v) for v in re.findall(r"handed on (\d+)", capfd.readouterr().err)]
        got[trips] = (st, offs, cols, rho, sc["list_rebuilds"], max(handed))
    assert got["40"][5] == 0 and got["1"][5] > 0 and got["0"][5] == 0
    assert got["40"][4] >= 2  # the lists were rebuilt along the way
    want = got["0"]
    for trips in ("40", "1"):
        st, offs, cols, rho, rebuilds, _ = got[trips]
        assert rebuilds == want[4]
        assert np.array_equal(offs, want[1]) and np.array_equal(cols, want[2])
        assert np.array_equal(rho.view(np.int32), want[3].view(np.int32))
        for k, a in st.items():
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(want[0][k]).view(np.uint8)), (trips, k)


@pytest.mark.parametrize("name", ["wcxsph_default", "wcxsph3d_block", "droplet3d"])
def test_neighbour_sets_from_lists_on_the_standing_order(name, monkeypatch):
    # The lists DensityReinit makes without a rebuild: the cell table is that of the stepping path's
    # last sort, the particles have moved on since (some into other columns of cells, where a
    # particle must still be kept out of its own row).  SPH_NEIGHBOURS_LISTS=standing lets
    # sph_neighbours read lists made that way: the sets must be the oracle's, frame after frame.
    monkeypatch.setenv("SPH_NEIGHBOURS_LISTS", "standing")
    spec = util.drop_cases()[name] if name in util.drop_cases() else util.small_cases()[name]
    orc, sim = start(spec)
    rebuilds = sim.scalars()["list_rebuilds"]
    probed_standing = 0
    for frame in range(15):
        orc.step(10)
        sc = sim.step(10)
        go, gc = sim.neighbours(capi.STATE_NP1)
        # the probe made no rebuild of its own when the stepping path's lists stood
        probed_standing += sim.scalars()["list_rebuilds"] == sc["list_rebuilds"]
        oo, occ = orc.neighbours(O.STATE_NP1)
        assert np.array_equal(go, oo) and np.array_equal(gc, occ), "frame %d" % frame
    assert probed_standing >= 12
    assert sim.scalars()["list_rebuilds"] - rebuilds >= 3  # the cells did go stale and were renewed along the way
    sim.close()


@pytest.mark.parametrize("name,frames,steps", [("wcxsph3d_block", 4, 7), ("wcxsph_fine", 4, 7), ("wcsph_fine", 4, 7),
                                               ("droplet3d", 6, 7), ("wcxsph_default", 14, 10)])
def test_density_reinit_on_the_standing_order_and_after_a_rebuild(name, frames, steps, monkeypatch):
    # After steps DensityReinit makes its lists on the order and cell table of the stepping path's
    # last rebuild (the particles have moved since; the build looks that much further); otherwise,
    # or with SPH_REINIT_LISTS=fresh, it rebuilds.  Same neighbour sets, so the same rho up to the
    # order of the sums: within one float ulp of each other and of the oracle, frame after frame.
    spec = util.drop_cases()[name] if name in util.drop_cases() else util.small_cases()[name]
    oc, _ = util.both_cases(spec)
    orc = O.Oracle(oc)
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    sims = {}
    for how in ("standing", "fresh"):
        monkeypatch.setenv("SPH_REINIT_LISTS", how)
        _, pc = util.both_cases(spec)
        sims[how] = capi.Sim.from_case(pc)
        sims[how].forces(capi.STATE_N)
        sims[how].zero_boundary_forces(capi.STATE_N)
    for frame in range(frames):
        orc.step(steps)
        orc.density_reinit()
        want = orc.get_state(O.STATE_NP1)["rho"]
        rho = {}
        for how, sim in sims.items():
            sc = sim.step(steps)
            sim.density_reinit()
            assert sim.reinit_info()["standing_lists"] == (how == "standing")
            rho[how] = sim.download(capi.STATE_NP1, names=("rho",))["rho"]
            assert util.float_ulp_diff(rho[how], want).max() <= 1, (how, frame)
        assert util.float_ulp_diff(rho["standing"], rho["fresh"]).max() <= 1
    # a call that follows an upload finds no standing lists and rebuilds
    sim = sims["standing"]
    sim.upload(capi.STATE_NP1, dict(rho=rho["standing"]))
    sim.density_reinit()
    assert not sim.reinit_info()["standing_lists"]
    for sim in sims.values():
        sim.close()


def test_density_reinit_in_one_launch_and_by_levels_give_the_same_bits(monkeypatch, capfd):
    # a single rank's large case is walked in one launch (every particle as soon as the neighbours
    # before it are done); slab runs walk level by level.  Same sums in the same order: same bits.
    # Third run: the one-launch walk is made to give up (its waves may wait 20 us for anything);
    # rho is put back and the walk is done by levels.
    spec = dict(program=2, npart=(64, 48, 24), pstep=0.1, box=(13.0, 5.1, 4.8), start=(0.15, 0.15, 0.15))
    rho = {}
    for walk, limit in (("flow", "30"), ("levels", "30"), ("flow", "2e-5")):
        monkeypatch.setenv("SPH_REINIT_WALK", walk)
        monkeypatch.setenv("SPH_REINIT_LIMIT_S", limit)
        _, pc = util.both_cases(spec)
        sim = capi.Sim.from_case(pc)
        sim.forces(capi.STATE_N)
        sim.zero_boundary_forces(capi.STATE_N)
        sim.step(30)
        for _ in range(2):
            sim.density_reinit()
        rho[walk, limit] = sim.download(capi.STATE_NP1)["rho"].copy()
        assert sim.npts > 100000
        sim.close()
    assert "gave up; walking by levels" in capfd.readouterr().err
    want = rho["levels", "30"].view(np.int32)
    assert np.array_equal(rho["flow", "30"].view(np.int32), want)
    assert np.array_equal(rho["flow", "2e-5"].view(np.int32), want)


@pytest.mark.parametrize("walk", ["flow", "levels"])
def test_density_reinit_with_rows_of_more_than_64_candidates(walk, monkeypatch):
    # a wavefront takes 64 candidates of a row at a time; lists with a wide skin make rows longer
    # than that (a second pass over the row, in the one-launch walk with its own waits)
    monkeypatch.setenv("SPH_ONESHOT_SKIN", "0.9")
    monkeypatch.setenv("SPH_REINIT_WALK", walk)
    spec = dict(program=2, npart=(24, 20, 16), pstep=0.1, box=(3.0, 2.4, 2.2), start=(0.15, 0.15, 0.15))
    orc, sim = start(spec)
    orc.step(6)
    sim.step(6)
    orc.density_reinit()
    sim.density_reinit()
    g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
    u = util.float_ulp_diff(g["rho"], o["rho"])
    print("rows beyond 64 (%s): rho max ulp %d, differing %d of %d" % (walk, u.max(), int((u > 0).sum()), len(u)))
    assert u.max() <= 1
    sim.close()


def test_density_reinit_on_a_uniform_lattice_keeps_rho0_inside():
    # known answer: MLS reproduces a constant field wherever the moment matrix is invertible
    _, pc = util.both_cases(CASES["wcxsph3d_block"])
    for prog in (0, 1, 2):
        sim = capi.Sim.from_case(capi.default_case(prog))
        sim.density_reinit()
        rho = sim.download(capi.STATE_NP1)["rho"]
        assert np.max(np.abs(rho.astype(np.float64) - 1000.0)) <= 1e-3, prog
        sim.close()
    sim = capi.Sim.from_case(pc)
    sim.density_reinit()
    rho = sim.download(capi.STATE_NP1)["rho"]
    assert np.max(np.abs(rho.astype(np.float64) - 1000.0)) <= 1e-3
    sim.close()


# ---- step lists (the second sweep of a step draws them from the candidates) ----------------------

@pytest.mark.parametrize("name", ["wcxsph3d_block", "wcxsph_fine", "wcsph_fine"])
def test_working_lists_give_the_same_answer(name):
    # the second sweep of each step writes the candidates within 1.025 R; the later sweeps and the
    # next step's first sweep walk those
    orc, sim = start(CASES[name], inner_skin_frac=0.025)
    orc.step(10)
    sc = sim.step(10)
    assert_state_close(sim.download(capi.STATE_N), orc.get_state(O.STATE_N), name)
    assert_scalars_close(sc, orc.scalars(), name)
    sim.close()


def test_working_lists_too_tight_fall_back_to_the_candidates():
    # a margin far smaller than the motion between a step's sweeps: every step is redone on the
    # full lists
    orc, sim = start(CASES["wcxsph3d_block"], inner_skin_frac=1e-13)
    orc.step(30)
    sim.step(29)
    before = sim.scalars()["step_redos"]
    sc = sim.step(1)
    assert sc["step_redos"] > before > 0
    assert_state_close(sim.download(capi.STATE_N), orc.get_state(O.STATE_N), "tight working lists")
    sim.close()


@pytest.mark.parametrize("skin,inner", [(0.3, 0.01), (1.5, 1.0)], ids=["narrow", "rows_past_the_staging_area"])
def test_step_lists_change_nothing_but_roundings(skin, inner):
    # rows of the step lists keep the candidates' order and only drop pairs outside the radius; the
    # launch that draws them is a different instantiation of the kernel (its compiler may fuse
    # multiply-adds differently), so fields agree to roundings, not bit for bit -- also with a wide
    # skin and many rebuilds.  With a draw radius of 2 R the rows of particles inside the fluid keep
    # more entries than the draw collects in LDS, so the remainder takes its path straight to memory.
    spec = CASES["wcxsph3d_block"]
    _, pc = util.both_cases(spec)
    out = []
    for inner_frac in (-1.0, inner):
        sim = capi.Sim.from_case(pc, skin_frac=skin, inner_skin_frac=inner_frac)
        sim.forces(capi.STATE_N)
        sim.zero_boundary_forces(capi.STATE_N)
        sc = sim.step(40)
        out.append((sc, sim.download(capi.STATE_NP1), sim.download(capi.STATE_N)))
        sim.close()
    (sa, pa, na), (sb, pb, nb) = out
    assert sa["hit_pairs"] == sb["hit_pairs"] and sa["step_redos"] == sb["step_redos"] == 0
    for k in ("dt", "t", "maxmu", "maxf", "errsum", "logbase"):
        assert abs(sa[k] - sb[k]) <= 1e-11 * abs(sa[k]), k
    for x, y in ((pa, pb), (na, nb)):
        for k in ("x", "v", "V", "f"):
            assert util.rel_err(x[k], y[k]) <= 1e-12, k
        assert util.float_ulp_diff(x["rho"], y["rho"]).max() <= 1


def test_every_row_names_particle_slots(monkeypatch):
    # the pair kernel fetches what a row names before it looks at the row's count, so every word of
    # every row -- entries, padding, leftovers of earlier builds -- must index the arrays; with
    # SPH_CHECK_LISTS the library verifies that after each list build (rows grown twice on the way)
    monkeypatch.setenv("SPH_CHECK_LISTS", "1")
    spec = CASES["wcxsph3d_block"]
    _, pc = util.both_cases(spec)
    sim = capi.Sim.from_case(pc, skin_frac=0.3, max_neighbours=16)
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    sc = sim.step(60)
    assert sc["step_redos"] == 0 and sc["hit_pairs"] > 0
    sim.density_reinit()
    sim.close()


# ---- droplet / jet bodies (initialisers beyond the reference's box block) ------------------------

@pytest.mark.parametrize("name", list(util.drop_cases()))
def test_droplet_and_jet_steps(name):
    spec = util.drop_cases()[name]
    orc, sim = start(spec)
    go, gc = sim.neighbours(capi.STATE_N)
    oo, occ = orc.neighbours(O.STATE_N)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    orc.step(20)
    sc = sim.step(20)
    assert_state_close(sim.download(capi.STATE_N), orc.get_state(O.STATE_N), name)
    assert_scalars_close(sc, orc.scalars(), name)
    assert sc["maxmu"] > 0  # the body has reached the pool or the wall: viscosity is at work
    sim.close()


# ---- edge cases ------------------------------------------------------------------------------------

def test_coincident_and_isolated_particles():
    # two fluid particles at one point (r = 0: kernel gradient 0, viscosity denominator 0.01 H^2) and
    # one far from everything (no neighbour but itself): finite, and as the oracle has it
    spec = CASES["wcxsph3d_block"]
    oc, pc = util.both_cases(spec)
    orc, sim = O.Oracle(oc), capi.Sim.from_case(pc)
    st = orc.get_state(O.STATE_N)
    x, v = st["x"].copy(), st["v"].copy()
    nb = orc.bound_parts
    x[nb + 1] = x[nb]                       # coincident pair
    v[nb] = (0.3, 0.0, -0.2)
    v[nb + 1] = (-0.1, 0.2, 0.0)
    x[nb + 7] = (1.7, 1.2, 1.05)            # alone in the corner of the box, above the block
    for w_o, w_g in ((O.STATE_N, capi.STATE_N), (O.STATE_NP1, capi.STATE_NP1)):
        orc.set_state(w_o, x=x, v=v)
        sim.upload(w_g, {"x": np.ascontiguousarray(x.T), "v": np.ascontiguousarray(v.T)})
    go, gc = sim.neighbours(capi.STATE_N)
    oo, occ = orc.neighbours(O.STATE_N)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    assert go[nb + 8] - go[nb + 7] == 1     # only itself
    orc.forces(O.STATE_N); orc.zero_boundary_f(O.STATE_N)
    sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
    g = sim.download(capi.STATE_N)
    assert all(np.isfinite(g[k]).all() for k in g)
    assert_state_close(g, orc.get_state(O.STATE_N), "coincident/isolated, Forces")
    orc.step(3)
    sc = sim.step(3)
    assert_state_close(sim.download(capi.STATE_N), orc.get_state(O.STATE_N), "coincident/isolated, 3 steps")
    assert_scalars_close(sc, orc.scalars(), "coincident/isolated")
    orc.density_reinit(); sim.density_reinit()
    assert util.float_ulp_diff(sim.download(capi.STATE_NP1)["rho"], orc.get_state(O.STATE_NP1)["rho"]).max() <= 1
    sim.close()


def test_smallest_cases():
    # one fluid particle in a box; and a 2-D column one particle wide
    for spec in (dict(program=2, npart=(1, 1, 1), pstep=0.1, box=(0.6, 0.6, 0.6), start=(0.3, 0.3, 0.3)),
                 dict(program=1, npart=(1, 7), pstep=0.1, box=(1.0, 1.5), start=(0.5, 0.12)),
                 dict(program=0, npart=(2, 2), pstep=0.1, box=(1.0, 1.0), start=(0.45, 0.11))):
        orc, sim = start(spec)
        orc.step(5)
        sc = sim.step(5)
        assert_state_close(sim.download(capi.STATE_N), orc.get_state(O.STATE_N), str(spec["npart"]))
        assert_scalars_close(sc, orc.scalars(), str(spec["npart"]))
        go, gc = sim.neighbours(capi.STATE_NP1)
        oo, occ = orc.neighbours(O.STATE_NP1)
        assert np.array_equal(go, oo) and np.array_equal(gc, occ)
        sim.close()


def test_a_particle_gone_nan_does_not_fault():
    # the reference only notices NaN when it writes a frame (WCXSPH3D.cpp:422-431); here a NaN position must not
    # take the neighbour search out of bounds: the step completes and the NaN is there to be found
    _, pc = util.both_cases(CASES["wcxsph3d_block"])
    sim = capi.Sim.from_case(pc)
    p, st = capi.case_init(pc)
    x = st["x"].copy()
    x[:, int(p.bound_parts) + 5] = np.nan
    sim.upload(capi.STATE_N, {"x": x})
    sim.upload(capi.STATE_NP1, {"x": x})
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    sim.step(2)
    g = sim.download(capi.STATE_NP1, names=("x",))["x"]
    assert np.isnan(g[:, int(p.bound_parts) + 5]).all()
    others = np.delete(g, int(p.bound_parts) + 5, axis=1)
    assert np.isfinite(others[:, : int(p.bound_parts)]).all()
    sim.close()


# ---- constants a settings file may change ---------------------------------------------------------
# rho_ref, viscosity, gravity (Sample_Input.txt:8-11) and the speed of sound replace the reference's
# compile-time constants (WCXSPH3D.cpp:54-56,142,195); the oracle takes the same overrides.

PHYS = dict(rho0=850.0, Cs=40.0, alpha=0.08, gravity=3.7)


@pytest.mark.parametrize("name", ["wcxsph3d_block", "wcxsph_fine", "wcsph_fine"])
def test_changed_physical_constants(name):
    spec = dict(CASES[name], phys=PHYS)
    orc, sim = start(spec)
    assert orc.k.rho0 == 850.0 and orc.k.g == -3.7
    assert_state_close(sim.download(capi.STATE_N), orc.get_state(O.STATE_N), name + " Forces")
    orc.step(10)
    sc = sim.step(10)
    assert_state_close(sim.download(capi.STATE_N), orc.get_state(O.STATE_N), name)
    assert_scalars_close(sc, orc.scalars(), name)
    go, gc = sim.neighbours(capi.STATE_NP1)
    oo, occ = orc.neighbours(O.STATE_NP1)
    assert np.array_equal(go, oo) and np.array_equal(gc, occ)
    orc.density_reinit()
    sim.density_reinit()
    assert util.float_ulp_diff(sim.download(capi.STATE_NP1)["rho"], orc.get_state(O.STATE_NP1)["rho"]).max() <= 1
    sim.close()
