"""Developer check: HIP path against the CPU oracle on the small cases, with diagnostics."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from oracle import oracle as O  # noqa: E402
from sph_amd import capi  # noqa: E402
import util  # noqa: E402


def cmp_state(tag, g, o, dim):
    for k in ("x", "v", "V", "f"):
        a, b = g[k].T, o[k]
        print("   %-5s %s max|d|=%.3e  rel=%.3e" % (k, tag, np.max(np.abs(a - b)), util.rel_err(a, b)))
    for k in ("rho", "p", "Rrho", "m"):
        u = util.float_ulp_diff(g[k], o[k])
        print("   %-5s %s max ulp=%d  nonzero=%d" % (k, tag, u.max(), int((u > 0).sum())))


def main():
    only = sys.argv[1:] or None
    for name, spec in util.small_cases().items():
        if only and name not in only:
            continue
        oc, pc = util.both_cases(spec)
        orc = O.Oracle(oc)
        sim = capi.Sim.from_case(pc)
        dim = sim.dim
        print("== %s  n=%d walls=%d" % (name, sim.npts, sim.bound_parts))
        # initial state equal?
        s0 = sim.download(capi.STATE_N)
        o0 = orc.get_state(O.STATE_N)
        assert np.array_equal(s0["x"].T, o0["x"]), "init x differs"
        assert np.array_equal(s0["m"], o0["m"]), "init m differs"
        # neighbour sets
        go, gc = sim.neighbours(capi.STATE_N)
        oo, occ = orc.neighbours(O.STATE_N)
        print("   neighbours: total %d vs %d  equal=%s" % (go[-1], oo[-1],
              np.array_equal(go, oo) and np.array_equal(gc, occ)))
        # t=0 forces
        sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
        orc.forces(O.STATE_N); orc.zero_boundary_f(O.STATE_N)
        cmp_state("t0", sim.download(capi.STATE_N), orc.get_state(O.STATE_N), dim)
        # sweeps
        for ns in (1, 2):
            sim2 = capi.Sim.from_case(pc)
            sim2.forces(capi.STATE_N); sim2.zero_boundary_forces(capi.STATE_N)
            orc2 = O.Oracle(oc)
            orc2.forces(O.STATE_N); orc2.zero_boundary_f(O.STATE_N)
            sc = sim2.sweeps(ns)
            orc2.step(1, ns)
            print("  sweeps=%d scalars gpu %s" % (ns, {k: sc[k] for k in ("errsum", "logbase", "maxmu")}))
            print("           scalars orc %s" % {k: orc2.scalars()[k] for k in ("errsum", "logbase", "maxmu")})
            cmp_state("np1", sim2.download(capi.STATE_NP1), orc2.get_state(O.STATE_NP1), dim)
            sim2.close()
        # steps
        t0 = time.time()
        sc = sim.step(10)
        tg = time.time() - t0
        t0 = time.time()
        orc.step(10)
        to = time.time() - t0
        print("  10 steps: gpu %.3fs oracle %.3fs" % (tg, to))
        print("   gpu", sc)
        print("   orc", orc.scalars())
        cmp_state("N", sim.download(capi.STATE_N), orc.get_state(O.STATE_N), dim)
        cmp_state("NP1", sim.download(capi.STATE_NP1), orc.get_state(O.STATE_NP1), dim)
        go, gc = sim.neighbours(capi.STATE_NP1)
        oo, occ = orc.neighbours(O.STATE_NP1)
        print("   neighbours after 10 steps: total %d vs %d  equal=%s" % (go[-1], oo[-1],
              np.array_equal(go, oo) and np.array_equal(gc, occ)))
        sim.close()


if __name__ == "__main__":
    main()
