"""Timing experiment: the pair launch of the benchmark case (lattice at rest) with a library built
with -include tools/pair_hooks.h -DSPH_ABLATE=n (1: every fetch aimed at the particle's own record, 2: the arithmetic of every
other round skipped, 3: both).  Results of such a library are wrong on purpose; only the launch
time is read.  SPH_AMD_LIB chooses the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from sph_amd import capi

sim = capi.Sim.from_case(B.dam_break_3d(*B.C3))
sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
sim.step(2)
sim.profile_enable(True); sim.profile_read()
sc = sim.step(6)
p = sim.profile_read()
ms, n = p["pair_ms_by_kind"], p["pair_launches_by_kind"]
print(os.environ.get("SPH_AMD_LIB", "default"), "pair launch on step lists %.3f ms (%d), on candidate lists %.3f ms (%d), rebuilds %d redos %d"
      % (ms[1] / max(n[1], 1), n[1], ms[0] / max(n[0], 1), n[0], p["rebuilds"], sc["step_redos"]), flush=True)
sim.close()
