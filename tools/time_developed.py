"""Development check: stepping rate of the benchmark case once the column is collapsing (lists are
rebuilt as the particles travel), next to the rate at rest that bench.py reports."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sph_amd import capi
skin = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
inner = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
sim = capi.Sim.from_case(bench.dam_break_3d(640, 320, 48), skin_frac=skin, inner_skin_frac=inner)
sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
done = 0
for target in [t for t in (20, 200, 600, 1200) if t <= last]:
    while done < target - 20:
        sim.step(20); done += 20
    sim.profile_enable(True); sim.profile_read()
    r0 = sim.scalars()
    t0 = time.perf_counter(); sc = sim.step(20); dt = time.perf_counter() - t0; done += 20
    pr = sim.profile_read()
    kinds = " ".join("%d x %.3f" % (n, m / max(n, 1)) for m, n in zip(pr["pair_ms_by_kind"], pr["pair_launches_by_kind"]))
    print("steps %4d t %.4f dt %.2e: %.2f ms/step, %.3g particle-steps/s, pair %.3f ms [on candidates / on step lists / draws: %s], rebuilds %d (%.1f ms each), redos %d, pairs/particle %.1f"
          % (done, sc["t"], sc["dt"], dt / 20 * 1e3, sim.npts * 20 / dt, pr["pair_ms"] / max(pr["pair_launches"], 1), kinds, pr["rebuilds"],
             pr["rebuild_ms"] / max(pr["rebuilds"], 1), sc["step_redos"] - r0["step_redos"], (sc["hit_pairs"] + sim.npts) / sim.npts), flush=True)
