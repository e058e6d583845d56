"""CPU tests of the oracle: known answers that follow from the reference source, agreement of
its two neighbour back-ends, and the committed golden vectors.

The reference holds no tests or fixtures (parity unpinned); the known answers below are the
ones its source fixes: particle counts, constants, t=0 forces and the first-sweep closed form.
"""
import math
import os

import numpy as np
import pytest

from oracle import oracle as O

import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- constants and counts (WCSPH.cpp:42-96, WCXSPH.cpp:36-92, WCXSPH3D.cpp:39-97) -----------

def test_counts_and_constants():
    k = O.derive(O.default_case(O.WCSPH))
    assert (k.dim, k.bound_parts, k.sim_pts, k.npts) == (2, 660, 950, 1610)
    assert k.simmass == 10.0 and k.boundmass == 10.0
    assert k.H == pytest.approx(0.09772050238058398, rel=0, abs=0)
    assert k.search_radius == 4 * k.HSQ + 0.001 * k.H

    k = O.derive(O.default_case(O.WCXSPH))
    assert (k.dim, k.bound_parts, k.sim_pts, k.npts) == (2, 800, 1920, 2720)
    assert k.simmass == 10.000000000000002 and np.float32(k.simmass) == np.float32(10.0)
    assert k.boundmass == 9.000000000000002 and np.float32(k.boundmass) == np.float32(9.0)
    assert k.H == 0.097720502380584
    assert k.search_radius == 4 * k.HSQ
    assert k.correc == 7 / (4 * math.pi * k.HSQ)

    k = O.derive(O.default_case(O.WCXSPH3D))
    assert (k.dim, k.bound_parts, k.sim_pts, k.npts) == (3, 2041, 125, 2166)
    assert k.simmass == 1.0 and k.boundmass == 1.0
    assert k.H == 0.0927058084855655
    assert k.correc == 524.3588498414309
    assert k.B == 357142.85714285716
    assert k.beta == 0.25 and k.subits == 8 and k.g == -9.81 and k.g_axis == 2


@pytest.mark.parametrize("variant,total,interior", [(O.WCSPH, 11492, 9), (O.WCXSPH, 23782, 9),
                                                     (O.WCXSPH3D, 96796, 27)])
def test_neighbour_totals_and_backends(variant, total, interior):
    orc = O.Oracle(O.default_case(variant))
    ob, cb = orc.neighbours(O.STATE_N, O.SEARCH_BRUTE)
    oc, cc = orc.neighbours(O.STATE_N, O.SEARCH_CELLS)
    assert int(ob[-1]) == total
    assert np.array_equal(ob, oc) and np.array_equal(cb, cc)
    counts = np.diff(ob.astype(np.int64))
    # an interior lattice particle sees itself and the full stencil
    assert interior in counts[orc.bound_parts:]
    # every row holds its own index, ascending
    for i in (0, orc.bound_parts, orc.npts - 1):
        row = cb[int(ob[i]):int(ob[i + 1])]
        assert i in row and np.all(np.diff(row.astype(np.int64)) > 0)


def test_backends_agree_on_jittered_positions():
    for variant in (O.WCXSPH, O.WCXSPH3D):
        case = O.default_case(variant)
        orc = O.Oracle(case)
        st = orc.get_state(O.STATE_N)
        orc.set_state(O.STATE_N, x=util.jitter(st["x"], case.pstep))
        ob, cb = orc.neighbours(O.STATE_N, O.SEARCH_BRUTE)
        oc, cc = orc.neighbours(O.STATE_N, O.SEARCH_CELLS)
        assert np.array_equal(ob, oc) and np.array_equal(cb, cc)


# ---- kernel (WCXSPH3D.cpp:121-137) -------------------------------------------------------------

def test_kernel_known_values():
    for variant in (O.WCXSPH, O.WCXSPH3D):
        orc = O.Oracle(O.default_case(variant))
        k = orc.k
        assert orc.w2kernel(0.0) == k.correc
        assert np.all(orc.w2gradk(np.zeros(k.dim), 0.0) == 0.0)
        assert orc.w2kernel(2.0 * k.H) == 0.0 and orc.w2kernel(3.0 * k.H) == 0.0
        r = np.zeros(k.dim)
        r[0] = 2.5 * k.H
        assert np.all(orc.w2gradk(r, 2.5 * k.H) == 0.0)
        # between: positive, decreasing
        a, b = orc.w2kernel(0.5 * k.H), orc.w2kernel(1.5 * k.H)
        assert a > b > 0


# ---- t=0 Forces and the first sweep (WCXSPH3D.cpp:456-458, 271-277) ----------------------------

@pytest.mark.parametrize("variant", [O.WCSPH, O.WCXSPH, O.WCXSPH3D])
def test_t0_forces_are_gravity(variant):
    orc = O.Oracle(O.default_case(variant))
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    st = orc.get_state(O.STATE_N)
    nb, dim = orc.bound_parts, orc.dim
    g = np.zeros(dim)
    g[dim - 1] = -9.81
    assert np.all(st["f"][:nb] == 0.0)
    assert np.all(st["f"][nb:] == g)
    assert np.all(st["Rrho"] == 0.0) and np.all(st["V"] == 0.0)
    assert orc.scalars()["maxmu"] == 0.0


@pytest.mark.parametrize("variant", [O.WCSPH, O.WCXSPH, O.WCXSPH3D])
def test_first_sweep_closed_form(variant):
    orc = O.Oracle(O.default_case(variant))
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    x0 = orc.get_state(O.STATE_N)["x"].copy()
    orc.step(1, 1)  # the first sweep only
    st = orc.get_state(O.STATE_NP1)
    nb, dim = orc.bound_parts, orc.dim
    dt, beta = 0.0002, np.float32(0.25)
    g = -9.81
    v = 0.0 + 0.5 * dt * (g + g)
    c1 = 0.5 * (dt * dt) * float(np.float32(1) - np.float32(2) * beta)
    c2 = dt * dt * float(beta)
    xg = x0[nb:, dim - 1] + dt * 0.0 + c1 * g + c2 * g
    assert np.all(st["v"][nb:, dim - 1] == v)
    assert np.all(st["v"][nb:, :dim - 1] == 0.0)
    assert np.all(st["x"][nb:, dim - 1] == xg)
    assert np.all(st["x"][:nb] == x0[:nb]) and np.all(st["v"][:nb] == 0.0)
    assert np.all(st["rho"] == np.float32(1000.0)) and np.all(st["p"] == 0.0)


def test_time_step_control():
    # after one step of free fall: dt = 0.3*min(sqrt(H/maxf), H/(Cs+maxmu))  (WCXSPH3D.cpp:293-298)
    orc = O.Oracle(O.default_case(O.WCSPH))
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    orc.step(1)
    sc, k = orc.scalars(), orc.k
    assert sc["maxf"] == 9.81 and sc["maxmu"] == 0.0
    assert sc["dt"] == 0.3 * min(math.sqrt(k.H / 9.81), k.H / (k.Cs + 0.0))
    assert sc["t"] == np.float32(sc["dt"])


def test_density_reinit_reproduces_constant_field():
    # MLS weights form a partition of unity, so with every rho = rho0 the filter returns rho0
    for variant in (O.WCSPH, O.WCXSPH, O.WCXSPH3D):
        orc = O.Oracle(O.default_case(variant))
        orc.density_reinit()
        rho = orc.get_state(O.STATE_NP1)["rho"]
        assert np.max(np.abs(rho.astype(np.float64) - 1000.0)) <= 1e-3


# ---- golden vectors -----------------------------------------------------------------------------

@pytest.mark.parametrize("name,variant", [("wcsph", O.WCSPH), ("wcxsph", O.WCXSPH), ("wcxsph3d", O.WCXSPH3D)])
def test_oracle_reproduces_golden(name, variant):
    gold = np.load(os.path.join(GOLDEN, "%s_default_3steps.npz" % name))
    orc = O.Oracle(O.default_case(variant))  # cell back-end; the fixtures were made with brute force
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    orc.step(int(gold["steps"]))
    st = orc.get_state(O.STATE_NP1)
    for k in ("x", "v", "V", "f", "rho", "p", "Rrho"):
        assert np.array_equal(st[k], gold[k]), k
    sc = orc.scalars()
    for k in ("dt", "maxmu", "maxf", "errsum", "logbase"):
        assert sc[k] == float(gold[k]), k
    assert np.float32(sc["t"]) == gold["t"]
    offs, cols = orc.neighbours(O.STATE_NP1)
    assert np.array_equal(offs, gold["nbr_offsets"]) and np.array_equal(cols, gold["nbr_cols"])


def test_changed_constants_follow_the_source_formulas():
    # B = rho0 Cs^2 / gam (WCXSPH3D.cpp:56); f = g at t = 0 with the changed gravity; the first
    # sweep's closed form with it (WCXSPH3D.cpp:271-277)
    c = O.make_case(O.WCXSPH3D, (6, 5, 4), 0.1, (1.2, 1.0, 0.9), (0.15, 0.15, 0.15),
                    phys=dict(rho0=850.0, Cs=40.0, alpha=0.08, gravity=3.7))
    o = O.Oracle(c)
    assert o.k.B == 850.0 * 40.0 ** 2 / 7.0 and o.k.alpha == 0.08 and o.k.g == -3.7
    o.forces(O.STATE_N)
    st = o.get_state(O.STATE_N)
    fluid = slice(o.bound_parts, None)
    assert np.all(st["rho"] == np.float32(850.0))
    assert np.all(st["f"][fluid, 2] == -3.7) and np.all(st["f"][fluid, :2] == 0.0)
    o.zero_boundary_f(O.STATE_N)
    x0 = st["x"].copy()
    o.step(1, 1)
    s1 = o.get_state(O.STATE_NP1)
    dt = 0.0002
    assert np.allclose(s1["v"][fluid, 2], -3.7 * dt, rtol=1e-14, atol=0)
    assert np.allclose(s1["x"][fluid, 2] - x0[fluid, 2], -0.5 * 3.7 * dt * dt, rtol=1e-9, atol=0)
