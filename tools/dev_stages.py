"""Developer check in stages (prints as it goes, so a GPU fault can be placed)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from sph_amd import capi  # noqa: E402
import util  # noqa: E402


def say(*a):
    print(*a, flush=True)


def main():
    stage = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "wcxsph3d_default"
    spec = util.small_cases()[name]
    oc, pc = util.both_cases(spec)
    say("stage", stage, name)
    sim = capi.Sim.from_case(pc)
    say(" created+uploaded n=%d" % sim.npts)
    orc = O.Oracle(oc)
    if stage == "nbr":
        go, gc = sim.neighbours(capi.STATE_N)
        oo, occ = orc.neighbours(O.STATE_N)
        say(" neighbours equal:", np.array_equal(go, oo) and np.array_equal(gc, occ), go[-1], oo[-1])
    elif stage == "forces":
        sim.forces(capi.STATE_N)
        say(" forces done")
        sim.zero_boundary_forces(capi.STATE_N)
        orc.forces(O.STATE_N)
        orc.zero_boundary_f(O.STATE_N)
        g, o = sim.download(capi.STATE_N), orc.get_state(O.STATE_N)
        for k in ("f", "V"):
            say("  %s max|d| %.3e" % (k, np.max(np.abs(g[k].T - o[k]))))
        say("  Rrho max|d| %.3e" % np.max(np.abs(g["Rrho"] - o["Rrho"])))
    elif stage == "sweeps":
        sim.forces(capi.STATE_N)
        sim.zero_boundary_forces(capi.STATE_N)
        orc.forces(O.STATE_N)
        orc.zero_boundary_f(O.STATE_N)
        for ns in (1, 2, 8):
            sc = sim.sweeps(ns)
            say(" sweeps", ns, "done", sc["errsum"], sc["maxmu"])
        orc.step(1, 8)
        g, o = sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1)
        for k in ("x", "v", "f", "V"):
            say("  %s rel %.3e" % (k, util.rel_err(g[k].T, o[k])))
        for k in ("rho", "p", "Rrho"):
            say("  %s max|d| %.3e of %.3e" % (k, np.max(np.abs(g[k].astype(np.float64) - o[k])), np.max(np.abs(o[k]))))
    elif stage == "step":
        sim.forces(capi.STATE_N)
        sim.zero_boundary_forces(capi.STATE_N)
        orc.forces(O.STATE_N)
        orc.zero_boundary_f(O.STATE_N)
        sc = sim.step(1)
        say(" step done", sc)
        orc.step(1)
        say(" oracle   ", orc.scalars())
        sc = sim.step(9)
        orc.step(9)
        say(" 10 steps", sc)
        say(" oracle  ", orc.scalars())
        for w, ow in ((capi.STATE_N, O.STATE_N), (capi.STATE_NP1, O.STATE_NP1)):
            g, o = sim.download(w), orc.get_state(ow)
            for k in ("x", "v", "f", "V"):
                say("  %s rel %.3e" % (k, util.rel_err(g[k].T, o[k])))
            for k in ("rho", "p", "Rrho"):
                say("  %s max|d| %.3e of %.3e" % (k, np.max(np.abs(g[k].astype(np.float64) - o[k])),
                                                  np.max(np.abs(o[k]))))
    sim.close()
    say(" closed")


if __name__ == "__main__":
    main()
