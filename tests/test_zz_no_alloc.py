"""The entry point that allocates nothing (sph_step_noalloc) against sph_step.

Same kernels, so the bar is bit-for-bit equality of every field and scalar.  Two calls in a
row on one GPU, the second on a smaller case: both use the same fixed scratch array, which
keeps the first call's contents."""
import numpy as np
import pytest

from sph_amd import capi

pytestmark = pytest.mark.gpu


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
