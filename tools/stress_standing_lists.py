"""Development stress of the lists made on the stepping path's standing order (what DensityReinit
uses between rebuilds): many steps of the small cases at several skins, the sets read through
sph_neighbours (SPH_NEIGHBOURS_LISTS=standing) against the oracle's every few steps, and rho after
DensityReinit against the oracle's.   python tools/stress_standing_lists.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SPH_NEIGHBOURS_LISTS"] = "standing"
import numpy as np
from oracle import oracle as O
from sph_amd import capi
import util

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cases = dict(util.small_cases()); cases.update(util.drop_cases())
bad = 0
for name, spec in cases.items():
    for skin in (0.0, 0.3, 0.5):
        oc, pc = util.both_cases(spec)
        orc = O.Oracle(oc); sim = capi.Sim.from_case(pc, skin_frac=skin)
        orc.forces(O.STATE_N); orc.zero_boundary_f(O.STATE_N); sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
        standing = probes = 0; worst = 0
        for k in range(steps // 5):
            orc.step(5); sc = sim.step(5)
            r0 = sc["list_rebuilds"]
            go, gc = sim.neighbours(capi.STATE_NP1)
            standing += sim.scalars()["list_rebuilds"] == r0; probes += 1
            oo, occ = orc.neighbours(O.STATE_NP1)
            if not (np.array_equal(go, oo) and np.array_equal(gc, occ)):
                print("  %s skin %.1f: sets differ after %d steps" % (name, skin, 5 * (k + 1)), flush=True); bad += 1; break
            if k % 2 == 1:
                orc.density_reinit(); sim.density_reinit()
                u = int(util.float_ulp_diff(sim.download(capi.STATE_NP1, names=("rho",))["rho"], orc.get_state(O.STATE_NP1)["rho"]).max())
                worst = max(worst, u)
                if u > 1:
                    print("  %s skin %.1f: rho %d ulp off after %d steps" % (name, skin, u, 5 * (k + 1)), flush=True); bad += 1; break
        print("%s skin %.1f: %d probes, %d on the standing order, %d rebuilds, rho worst %d ulp, redos %d"
              % (name, skin, probes, standing, sc["list_rebuilds"], worst, sc["step_redos"]), flush=True)
        sim.close()
print("FAILED" if bad else "all equal")
sys.exit(1 if bad else 0)
