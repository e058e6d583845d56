"""Development check: a long run of a shipped case (frames of 10 steps + DensityReinit) on the GPU
and in the oracle, reporting drift, list rebuilds and step redos along the way."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from sph_amd import capi
import util

prog = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
orc = O.Oracle(O.default_case(prog)); sim = capi.Sim.from_case(capi.default_case(prog))
orc.forces(O.STATE_N); orc.zero_boundary_f(O.STATE_N); sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
t0 = time.time()
for fr in range(1, frames + 1):
    orc.step(10); sc = sim.step(10); orc.density_reinit(); sim.density_reinit()
    if fr % max(frames // 10, 1) == 0:
        g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
        go, gc = sim.neighbours(capi.STATE_NP1); oo, oc_ = orc.neighbours(O.STATE_NP1)
        print("frame %4d t %.4f dt %.3e  x %.2e v %.2e rho-ulp %d  sets equal %s  rebuilds %d redos %d  vmax %.2f  (%.0f s)"
              % (fr, sc["t"], sc["dt"], util.rel_err(g["x"].T, o["x"]), util.rel_err(g["v"].T, o["v"]),
                 int(util.float_ulp_diff(g["rho"], o["rho"]).max()), bool(np.array_equal(go, oo) and np.array_equal(gc, oc_)),
                 sc["list_rebuilds"], sc["step_redos"], float(np.abs(g["v"]).max()), time.time() - t0), flush=True)
