"""Development check: stepping rate of the 2-D single-mass case at ~1 M particles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi
case = capi.make_case(capi.PROGRAM_WCSPH, (833, 1267), 0.003, (15.0, 8.0), (0.31, 0.31))
p, _, _ = capi.case_params(case)
case.dt0 = 0.3 * p.H / p.Cs
sim = capi.Sim.from_case(case)
sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
sim.step(5)
sim.profile_enable(True); sim.profile_read()
t0 = time.perf_counter(); sc = sim.step(50); dt = time.perf_counter() - t0
pr = sim.profile_read()
print("n %d: %.3f ms/step, %.3g particle-steps/s, pair kernel %.3f ms, rebuilds %d, redos %d, pairs/particle %.1f"
      % (sim.npts, dt / 50 * 1e3, sim.npts * 50 / dt, pr["pair_ms"] / max(pr["pair_launches"], 1), pr["rebuilds"],
         sc["step_redos"], (sc["hit_pairs"] + sim.npts) / sim.npts))
