"""The entry point that allocates nothing (sph_step_noalloc) against sph_step.

Same kernels, so the bar is bit-for-bit equality of every field and scalar.  Two calls in a
row on one GPU, the second on a smaller case: both use the same fixed scratch array, which
keeps the first call's contents."""
import ctypes

import numpy as np
import pytest

from sph_amd import capi

pytestmark = pytest.mark.gpu


def _free_device_bytes():
    """hipMemGetInfo of the current device, from the runtime the library itself is linked to."""
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _run(case, nsteps, noalloc):
    sim = capi.Sim.from_case(case)
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    sim.step(1)  # the first rebuild sizes whatever grows on demand
    sc = sim.step_noalloc(nsteps) if noalloc else sim.step(nsteps)
    st = sim.download(capi.STATE_NP1)
    stn = sim.download(capi.STATE_N)
    sim.close()
    return sc, st, stn


def _same(a, b):
    sa, pa, na = a
    sb, pb, nb = b
    for k in ("dt", "t", "maxmu", "maxf", "errsum", "logbase", "hit_pairs", "list_rebuilds", "step_redos"):
        assert sa[k] == sb[k], k
    for x, y in ((pa, pb), (na, nb)):
        for k in x:
            assert np.array_equal(x[k].view(np.uint8), y[k].view(np.uint8)), k


def test_two_calls_in_a_row_the_second_smaller():
    big = capi.make_case(capi.PROGRAM_WCXSPH3D, (40, 32, 24), 0.1, (8.3, 3.5, 4.8), (0.15, 0.15, 0.15))
    small = capi.default_case(capi.PROGRAM_WCXSPH)
    ref_big = _run(big, 6, False)
    ref_small = _run(small, 12, False)
    got_big = _run(big, 6, True)
    got_small = _run(small, 12, True)
    _same(got_big, ref_big)
    _same(got_small, ref_small)
    assert ref_big[0]["hit_pairs"] > 0 and ref_small[0]["hit_pairs"] > 0


def test_sizes_beyond_the_scratch_array_are_refused():
    nmax = 4194304  # SPH_NOALLOC_MAX_PARTICLES
    big = capi.make_case(capi.PROGRAM_WCXSPH3D, (200, 160, 140), 0.1, (40.3, 16.3, 28.0), (0.15, 0.15, 0.15))
    pb, _, _ = capi.case_params(big)
    assert int(pb.npts) > nmax
    sim = capi.Sim.from_case(big)
    with pytest.raises(capi.SphError, match="scratch"):
        sim.step_noalloc(1)
    sim.close()


def test_the_call_leaves_the_free_device_memory_as_it_found_it():
    sim = capi.Sim.from_case(capi.make_case(capi.PROGRAM_WCXSPH3D, (40, 32, 24), 0.1, (8.3, 3.5, 4.8), (0.15, 0.15, 0.15)))
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    sim.step(2)  # rebuilds, draws and event pools have their sizes now
    before = _free_device_bytes()
    sc = sim.step_noalloc(8)
    assert _free_device_bytes() == before
    assert sc["list_rebuilds"] >= 1
    sim.close()


def test_rows_that_overflow_end_the_call_and_sph_step_makes_the_step_again():
    """Rows of four entries cannot hold a step's candidate lists: the call has to say so without
    allocating, leave state n alone, and the same step through sph_step gives what a context that
    never went through the refused call gives."""
    case = capi.make_case(capi.PROGRAM_WCXSPH3D, (12, 10, 8), 0.1, (2.3, 1.5, 1.8), (0.15, 0.15, 0.15))

    def fresh():
        sim = capi.Sim.from_case(case, max_neighbours=4)
        return sim

    ref = fresh()
    sc_ref = ref.step(2)
    want_np1, want_n = ref.download(capi.STATE_NP1), ref.download(capi.STATE_N)
    ref.close()

    sim = fresh()
    n_before = sim.download(capi.STATE_N)
    free_before = _free_device_bytes()
    with pytest.raises(capi.SphError) as err:
        sim.step_noalloc(2)
    assert err.value.code == capi.ENOMEM
    assert _free_device_bytes() == free_before
    n_after = sim.download(capi.STATE_N)
    for k in n_before:
        assert np.array_equal(n_before[k].view(np.uint8), n_after[k].view(np.uint8)), k
    sc = sim.step(2)
    got_np1, got_n = sim.download(capi.STATE_NP1), sim.download(capi.STATE_N)
    sim.close()
    for k in ("dt", "t", "maxmu", "maxf", "errsum", "logbase", "hit_pairs"):
        assert sc[k] == sc_ref[k], k
    for x, y in ((got_np1, want_np1), (got_n, want_n)):
        for k in x:
            assert np.array_equal(x[k].view(np.uint8), y[k].view(np.uint8)), k
