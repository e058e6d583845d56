"""Development timing of the list rebuild at C3 with a fixed skin: the lattice at rest, two steps,
every one of them forced to rebuild (lists for one use), times from the library's profile records.
    python tools/time_build.py [skin]     (SPH_AMD_LIB chooses the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sph_amd import capi

skin = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
case = bench.dam_break_3d(640, 320, 48)
sim = capi.Sim.from_case(case, skin_frac=skin)
sim.forces(capi.STATE_N)
sim.zero_boundary_forces(capi.STATE_N)
sim.profile_enable(True)
for k in range(3):
    sim.step(1)
    # touching the state invalidates the lists: the next step rebuilds
    sim.upload(capi.STATE_NP1, dict(rho=sim.download(capi.STATE_NP1, names=("rho",))["rho"]))
pr = sim.profile_read()
print("skin %.2f: %d rebuilds, %.3f ms each; pair %.3f ms" % (skin, pr["rebuilds"], pr["rebuild_ms"] / max(pr["rebuilds"], 1), pr["pair_ms"] / max(pr["pair_launches"], 1)))
sim.close()
