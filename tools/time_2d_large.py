"""Development check: the pair kernel on 2-D cases far larger than the on-die caches, where the
path is bound by memory traffic: algorithmic bytes per launch over the kernel's launch time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi

PEAK = 8.0e12
# algorithmic bytes per particle and sweep of the fused launch: pair stage (60 / 80 B in 2-D) plus
# the update stage (148 B in 2-D with XSPH, 132 B without the V arrays)
for prog, name, bytes_pp in ((capi.PROGRAM_WCSPH, "WCSPH", 60 + 132), (capi.PROGRAM_WCXSPH, "WCXSPH", 80 + 148)):
    case = capi.make_case(prog, (4800, 6700), 0.0005, (15.0, 8.0), (0.31, 0.31))
    p, _, _ = capi.case_params(case)
    case.dt0 = 0.3 * p.H / p.Cs
    sim = capi.Sim.from_case(case)
    sim.forces(capi.STATE_N); sim.zero_boundary_forces(capi.STATE_N)
    sim.step(3)
    sim.profile_enable(True); sim.profile_read()
    t0 = time.perf_counter(); sc = sim.step(10); dt = time.perf_counter() - t0
    pr = sim.profile_read()
    pair = pr["pair_ms"] / max(pr["pair_launches"], 1) * 1e-3
    print("%s n %d: %.2f ms/step, %.3g particle-steps/s; pair kernel %.3f ms = %.2f TB/s of %d B/particle (%.1f%% of 8 TB/s); "
          "rebuilds %d, pairs/particle %.1f" % (name, sim.npts, dt / 10 * 1e3, sim.npts * 10 / dt, pair * 1e3,
          bytes_pp * sim.npts / pair / 1e12, bytes_pp, 100 * bytes_pp * sim.npts / pair / PEAK, pr["rebuilds"],
          (sc["hit_pairs"] + sim.npts) / sim.npts), flush=True)
    sim.close()
