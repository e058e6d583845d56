"""Development check: the three programs as shipped -- time per step and per DensityReinit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi
for prog, name in ((0, "WCSPH"), (1, "WCXSPH"), (2, "WCXSPH3D")):
    sim = capi.Sim.from_case(capi.default_case(prog))
    sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
    sim.step(50)
    t0 = time.perf_counter(); sim.step(500); t1 = time.perf_counter()
    for _ in range(5): sim.density_reinit()
    t2 = time.perf_counter()
    print("%-9s n %5d: %.3f ms/step, DensityReinit %.2f ms" % (name, sim.npts, (t1 - t0) / 500 * 1e3, (t2 - t1) / 5 * 1e3), flush=True)
    sim.close()
