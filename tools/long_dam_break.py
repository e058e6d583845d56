"""Development check: the 3-D dam break far into the flow -- frames of 10 steps + DensityReinit, as
the reference's main loop runs them -- reporting rate, list rebuilds, step redos and whether every
field stays finite.   usage: python tools/long_dam_break.py NXxNYxNZ frames"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sph_amd import capi

nx, ny, nz = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "160x96x48").split("x"))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sim = capi.Sim.from_case(bench.dam_break_3d(nx, ny, nz))
sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
n = sim.npts
t0 = time.perf_counter(); t_re = 0.0
for fr in range(1, frames + 1):
    sc = sim.step(10)
    a = time.perf_counter(); sim.density_reinit(); t_re += time.perf_counter() - a
    if fr % max(frames // 10, 1) == 0:
        st = sim.download(capi.STATE_NP1, names=("x", "v", "rho"))
        ok = all(np.isfinite(st[k]).all() for k in st)
        el = time.perf_counter() - t0
        print("frame %4d t %.3f dt %.2e: %.0f M particle-steps/s overall (reinit %.1f ms each), rebuilds %d redos %d, |v|max %.2f, rho %.1f..%.1f, finite %s"
              % (fr, sc["t"], sc["dt"], n * fr * 10 / el / 1e6, t_re / fr * 1e3, sc["list_rebuilds"], sc["step_redos"],
                 float(np.abs(st["v"]).max()), float(st["rho"].min()), float(st["rho"].max()), ok), flush=True)
        if not ok:
            sys.exit(1)
sim.close()
