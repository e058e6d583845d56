"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

The reference ships no fixtures and cannot be built here (it needs Eigen3 and nanoflann), so
these vectors pin the ORACLE against regressions; they are not outputs of the reference
("parity unpinned").  Inputs are the three programs' shipped constants; the expected outputs
are the state n+1 and the run scalars after the t=0 Forces call and 3 Newmark-beta steps.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

NAMES = {O.WCSPH: "wcsph", O.WCXSPH: "wcxsph", O.WCXSPH3D: "wcxsph3d"}
STEPS = 3


def main():
    for variant, name in NAMES.items():
        orc = O.Oracle(O.default_case(variant), search=O.SEARCH_BRUTE)
        orc.forces(O.STATE_N)
        orc.zero_boundary_f(O.STATE_N)
        orc.step(STEPS)
        st = orc.get_state(O.STATE_NP1)
        sc = orc.scalars()
        offs, cols = orc.neighbours(O.STATE_NP1, O.SEARCH_BRUTE)
        np.savez_compressed(
            os.path.join(HERE, "%s_default_%dsteps.npz" % (name, STEPS)),
            steps=STEPS, npts=orc.npts, bound_parts=orc.bound_parts,
            x=st["x"], v=st["v"], V=st["V"], f=st["f"], rho=st["rho"], p=st["p"], Rrho=st["Rrho"],
            dt=sc["dt"], t=np.float32(sc["t"]), maxmu=sc["maxmu"], maxf=sc["maxf"],
            errsum=sc["errsum"], logbase=sc["logbase"],
            nbr_offsets=offs, nbr_cols=cols)
        print(name, orc.npts, "pairs", int(offs[-1]), "dt", sc["dt"])


if __name__ == "__main__":
    main()
