"""Development aid: per-phase cycles of a fluid wave of the pair kernel.  Needs a library built
with -DSPH_PHASE_TIMING (make -C sph_amd/csrc -B HIPFLAGS="... -DSPH_PHASE_TIMING")."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sph_amd import capi
sim = capi.Sim.from_case(bench.dam_break_3d(640, 320, 48))
sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
sim.step(2)
L = capi.lib()
buf = (C.c_ulonglong * 8)()
L.sph_debug_read_phases(buf, 1)
sim.step(3)
L.sph_debug_read_phases(buf, 1)
n = max(buf[7], 1)
print("waves sampled %d, mean rounds %.2f" % (buf[7], buf[4] / n))
for k, name in enumerate(["prologue (entry -> loop)", "pair loop", "update (loop end -> before reductions)", "reductions + barrier + partials"]):
    print("  %-42s %8.0f cycles" % (name, buf[k] / n))
