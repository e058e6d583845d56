"""Development check: the oracle against itself with ONE coordinate moved by one unit in the last
place, through frames of 10 steps + DensityReinit of the shipped 2-D WCXSPH case.  Shows how fast
the flow amplifies a rounding-size difference, i.e. what drift any re-ordering of sums must show."""
import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from oracle import oracle as O
import util
a=O.Oracle(O.default_case(1)); b=O.Oracle(O.default_case(1))
st=b.get_state(O.STATE_N); x=st["x"].copy(); i=a.bound_parts+777; x[i,0]=np.nextafter(x[i,0],10.0)
b.set_state(O.STATE_N,x=x); b.set_state(O.STATE_NP1,x=x)
for s in (a,b): s.forces(O.STATE_N); s.zero_boundary_f(O.STATE_N)
for fr in range(1,181):
    for s in (a,b): s.step(10); s.density_reinit()
    if fr%30==0:
        pa,pb=a.get_state(O.STATE_NP1),b.get_state(O.STATE_NP1)
        print(fr, "x %.2e v %.2e"%(util.rel_err(pa["x"],pb["x"]),util.rel_err(pa["v"],pb["v"])),flush=True)
