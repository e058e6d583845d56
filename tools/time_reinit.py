"""Wall time of DensityReinit and of a frame (10 steps + DensityReinit) at the benchmark's size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sph_amd import capi

nx, ny, nz = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "640x320x48").split("x"))
sim = capi.Sim.from_case(bench.dam_break_3d(nx, ny, nz))
sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
sim.step(2)
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # steps advanced first (developed flow)
while pre > 0:
    sim.step(min(pre, 50))
    pre -= 50
for rep in range(2):
    t0 = time.perf_counter(); sim.step(10); t1 = time.perf_counter(); sim.density_reinit(); t2 = time.perf_counter()
    print("n %d: 10 steps %.1f ms, DensityReinit %.1f ms" % (sim.npts, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
