"""Development timing of the reference's frame loop at C3 (10 steps + DensityReinit) in developed
flow, with what the frames spend where.   python tools/time_frames.py [frames] [preroll steps]
(SPH_REINIT_LISTS=fresh: DensityReinit rebuilds the lists instead of using the standing order)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sph_amd import capi

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 600
sim = capi.Sim.from_case(bench.dam_break_3d(640, 320, 48))
sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
for _ in range(pre // 10):
    sim.step(10); sim.density_reinit()
sim.profile_enable(True); sim.profile_read()
r0 = sim.scalars()["list_rebuilds"]
t0 = time.perf_counter(); t_re = 0.0
for _ in range(frames):
    sim.step(10)
    a = time.perf_counter(); sim.density_reinit(); t_re += time.perf_counter() - a
el = time.perf_counter() - t0
p = sim.profile_read()
ms, n = p["pair_ms_by_kind"], p["pair_launches_by_kind"]
print("%s lists: frame %.1f ms (reinit %.1f ms); per frame: pair launches on candidates %.1f x %.3f ms, on step lists %.1f x %.3f ms, "
      "draws %.1f x %.3f ms, rebuilds %.2f (timed ones %.2f x %.2f ms)"
      % (os.environ.get("SPH_REINIT_LISTS", "standing"), el / frames * 1e3, t_re / frames * 1e3, n[0] / frames, ms[0] / max(n[0], 1),
         n[1] / frames, ms[1] / max(n[1], 1), n[2] / frames, ms[2] / max(n[2], 1),
         (sim.scalars()["list_rebuilds"] - r0) / frames, p["rebuilds"] / frames, p["rebuild_ms"] / max(p["rebuilds"], 1)))
sim.close()
