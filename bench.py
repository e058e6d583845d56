"""Headline benchmark: particle-steps/s of the 3-D WCXSPH dam break.

One "step" is one full Newmark-beta call of the reference (WCXSPH3D.cpp:251-303): eight
sweeps, each a fixed-radius neighbour search + Forces + update.  The reference rebuilds its
search index before every sweep (WCXSPH3D.cpp:278); here candidate lists are rebuilt when the
particles have travelled far enough to need it, which never happens while the column is still
at rest.  The headline is therefore measured in DEVELOPED flow: the case is first advanced
`--preroll` steps (untimed set-up of the workload, like filling the arrays), then W warm-up
steps, then exactly K timed steps with list rebuilds among them.  The rate of the resting
lattice (no rebuilds) is reported beside it as `config.at_rest`.  Inputs are resident in HBM
before any timed region.  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... [--scaling strong]
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X peaks used for the roofline fractions
HBM_PEAK_GBS = 8000.0      # HBM3E spec
FP64_PEAK_TFLOPS = 78.6    # FP64 vector spec
# Algorithmic traffic per particle, 3-D, reference precisions.  Fused pair+update kernel, per
# sweep: pair stage 112 B (read x,v,rho,p,m 60 B; write f,V,Rrho 52 B) + integrate/EOS stage
# 212 B.  Neighbour build, per rebuild: read x 24 B, key+index 8 B, four radix passes of 16 B,
# cell table 8 B = 104 B.  Algorithmic flops: 80 per pair inside the radius (self included).
B_ALG_PAIR = 112.0
B_ALG_UPDATE = 212.0
# In a single domain the last of a step's eight sweeps also writes state n (pn = pnp1): x, v, f, V
# 96 B and rho, p, Rrho 12 B per particle, which a slab run's separate commit kernel writes instead.
B_ALG_COMMIT = 108.0
SWEEPS_PER_STEP = 8
B_ALG_REBUILD = 104.0
FLOP_PER_PAIR = 80.0

C3 = (640, 320, 48)
C4 = (2048, 1024, 48)


def dam_break_3d(nx, ny, nz):
    """3-D WCXSPH3D physics, Pstep 0.1 kept (H does not scale with Pstep in the 3-D program),
    fluid block nx*ny*nz in a wide shallow box twice as long as the block."""
    from sph_amd import capi
    pstep = 0.1
    start = (0.15, 0.15, 0.15)
    box = (2 * nx * pstep + 0.3, ny * pstep + 0.3, 2 * nz * pstep)
    return capi.make_case(capi.PROGRAM_WCXSPH3D, (nx, ny, nz), pstep, box, start)


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("sph_kernels.h", "sph_types.h"):
        with open(os.path.join(ROOT, "sph_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_counters(fluid, n):
    """The committed counter run of this same workload and of THIS kernel source (rocprofv3 cannot
    run inside this process), or None when there is none that matches."""
    try:
        with open(os.path.join(ROOT, "profiles", "pair_traffic.json")) as fh:
            t = json.load(fh)
    except OSError:
        return None
    if list(t["workload"]["fluid"]) != list(fluid) or t["particles"] != n:
        return None
    if t.get("kernel_source_sha256_16") != kernel_source_hash():
        return None
    return t


def measured_traffic(fluid, n):
    """HBM bytes per launch of the pair kernel: FETCH_SIZE with its gfx950 correction + WRITE_SIZE."""
    t = committed_counters(fluid, n)
    if t is None:
        return None
    reads = t["fetch_size_kib"] * 1024.0 + 0.5 * t["streamed_read_bytes_per_particle"] * n
    return reads + t["write_size_kib"] * 1024.0


def measured_issue(fluid, n):
    """What the pair launch keeps busy, from the same counter run: share of the SIMD-cycles in
    which a vector instruction issues, share of the launch the address units are busy, vector
    instructions per wave, share of their life the waves spend at a wait instruction."""
    t = committed_counters(fluid, n)
    return t.get("issue") if t else None


def cpu_baseline(sample=(64, 48, 32), steps=4):
    """The oracle (a port: the reference itself needs Eigen+nanoflann) on one host core: the
    2-D WCXSPH program as shipped for 20 steps, and a reduced block of the benchmark's 3-D
    case.  `value` is the 3-D figure.  (A 1 M-particle block would take ~80 s of CPU for five
    steps; the leg is held to ~15 s so that the default run finishes in about a minute.)"""
    from oracle import oracle as O
    orc = O.Oracle(O.default_case(O.WCXSPH))
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    t0 = time.perf_counter()
    orc.step(20)
    dt1 = time.perf_counter() - t0
    n1 = orc.npts
    orc.close()
    pc = dam_break_3d(*sample)
    oc = O.make_case(O.WCXSPH3D, tuple(pc.npart), pc.pstep, tuple(pc.box), tuple(pc.start))
    orc = O.Oracle(oc)
    orc.forces(O.STATE_N)
    orc.zero_boundary_f(O.STATE_N)
    t0 = time.perf_counter()
    orc.step(steps)
    dt = time.perf_counter() - t0
    n = orc.npts
    orc.close()
    return {
        "value": n * steps / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
        "sample": "CPU restatement of jammy4536/SPH (original needs Eigen+nanoflann), 1 core: "
                  "3-D dam break, fluid %dx%dx%d, %d particles, %d step(s), %.1f s; the 2-D WCXSPH "
                  "program as shipped (%d particles), 20 steps, %.2f s = %.0f particle-steps/s"
                  % (sample[0], sample[1], sample[2], n, steps, dt, n1, dt1, n1 * 20 / dt1),
    }


class Legs:
    """The timed legs of one rank; `sync` brackets a timed region (device sync, and a barrier
    over the ranks when there are several), `agree_max` folds a list of floats over the ranks."""

    def __init__(self, sim, args, sync, agree_max):
        self.sim, self.args, self.sync, self.agree_max = sim, args, sync, agree_max

    def timed(self, nsteps):
        sim = self.sim
        sim.profile_enable(True)
        sim.profile_read()
        r0 = sim.scalars()
        self.sync()
        t0 = time.perf_counter()
        sc = sim.step(nsteps)
        self.sync()
        elapsed = time.perf_counter() - t0
        prof = sim.profile_read()
        sim.profile_enable(False)
        launches = max(prof["pair_launches"], 1)
        elapsed, pair_ms, rb_ms, dr_ms = self.agree_max([
            elapsed, prof["pair_ms"] / launches, prof["rebuild_ms"] / max(prof["rebuilds"], 1),
            prof["pair_ms_by_kind"][2] / max(prof["pair_launches_by_kind"][2], 1)])
        cm, cn = prof["comm_ms"], prof["comm_counts"]
        comm = self.agree_max([cm[k] / max(cn[k], 1) for k in range(4)] + [float(c) for c in cn])
        return {
            "comm": {
                "what": "slowest rank: a sweep's halo exchange (HIP events on the communication stream; it runs beside the "
                        "interior pair launch, whose time pair_ms includes the wait for it), the step's all-reduce of its "
                        "scalars (events on the compute stream), and per list rebuild the halo set-up and the migration "
                        "(host time, both wait for the device)",
                "halo_exchange_ms_per_sweep": comm[0], "allreduce_ms_per_step": comm[1],
                "halo_setup_ms_per_rebuild": comm[2], "migration_ms_per_rebuild": comm[3],
                "halo_exchanges": int(comm[4]), "allreduces": int(comm[5]), "halo_setups": int(comm[6]),
                "migrations": int(comm[7]),
            },
            "elapsed": elapsed, "steps": nsteps, "ms_per_step": elapsed / nsteps * 1e3,
            "pair_ms": pair_ms, "pair_launches": prof["pair_launches"],
            "launches_on_step_lists": prof["pair_launches_by_kind"][1],
            "rebuilds": prof["rebuilds"], "rebuild_ms_each": rb_ms,
            "draws": prof["pair_launches_by_kind"][2], "draw_ms_each": dr_ms,
            "step_redos": sc["step_redos"] - r0["step_redos"], "hit_pairs": sc["hit_pairs"],
            "t": sc["t"], "dt": sc["dt"],
        }

    def run(self):
        """At rest: W warm-up + K_rest timed steps from t = 0.  Then the pre-roll, W warm-up
        steps and exactly K timed steps in developed flow, then one DensityReinit."""
        a, sim = self.args, self.sim
        done = 0
        for _ in range(a.warmup):
            sim.step(1)
            done += 1
        k_rest = min(a.steps, 10)
        rest = self.timed(k_rest)
        done += k_rest
        while done < a.preroll:
            m = min(20, a.preroll - done)
            sim.step(m)
            done += m
        for _ in range(a.warmup):
            sim.step(1)
        dev = self.timed(a.steps)
        self.sync()
        t0 = time.perf_counter()
        sim.density_reinit()
        self.sync()
        reinit_ms = self.agree_max([(time.perf_counter() - t0) * 1e3])[0]
        return rest, dev, reinit_ms


def pair_bytes_per_particle(single_domain):
    """Algorithmic bytes per particle of the pair launch, averaged over a step's sweeps."""
    return B_ALG_PAIR + B_ALG_UPDATE + (B_ALG_COMMIT / SWEEPS_PER_STEP if single_domain else 0.0)


def rooflines(n_launch, n_total, dev, pairs_per_particle):
    """`roofline` of the dominant kernel (the fused pair/update launch of a sweep) and of the
    list rebuild, from the launch durations measured with HIP events in the timed region.
    n_launch: particles one launch covers (the slowest rank's in a slab run)."""
    per_particle = pair_bytes_per_particle(n_launch == n_total)
    b_alg = per_particle * n_launch
    ach = b_alg / (dev["pair_ms"] * 1e-3) / 1e9
    f_alg = FLOP_PER_PAIR * pairs_per_particle * n_launch
    ach_tf = f_alg / (dev["pair_ms"] * 1e-3) / 1e12
    pair = {
        "kernel": "k_pair<3,xsph,fused> (Forces + wall reset + Newmark update + EOS, one launch per sweep)",
        # achieved/peak/frac price the launch against the HBM roof, as the contract asks; `bound` names
        # what the launch is held by, which is neither of the two roofs
        "bound": "latency", "roof": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
        "traffic": None, "issue": None, "algorithmic_bytes_per_launch": b_alg, "kernel_ms": dev["pair_ms"],
        "launches": dev["pair_launches"], "algorithmic_bytes_per_particle_sweep": per_particle,
        "limiter": "neither roof binds: a round of the pair loop is a chain -- the quads' gathers land in LDS, are read back transposed, then ~100 "
                   "FP64 instructions -- and a wave has six of them per 16 particles with one round of look-ahead: the waves "
                   "spend over half their life at a wait instruction while the vector ALU is ~2/3 busy and the address units "
                   "~60 %; with every gather aimed at the particle's own record the launch takes 14 % less, with half the "
                   "arithmetic 3 % less, with both 21 % less (tools/ablate_pair.py)",
        "fp64": {"achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": ach_tf / FP64_PEAK_TFLOPS, "flop_per_pair": FLOP_PER_PAIR},
    }
    rb = None
    if dev["rebuilds"]:
        b_rb = B_ALG_REBUILD * n_launch
        a_rb = b_rb / (dev["rebuild_ms_each"] * 1e-3) / 1e9
        rb = {
            "kernel": "list rebuild (bbox, hash, radix sort, reorder, cell table, k_build_list dominant)",
            "bound": "hbm", "achieved": a_rb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_rb / HBM_PEAK_GBS,
            "traffic": None, "algorithmic_bytes_per_rebuild": b_rb, "rebuild_ms": dev["rebuild_ms_each"],
            "rebuilds": dev["rebuilds"], "algorithmic_bytes_per_particle": B_ALG_REBUILD,
        }
    return pair, rb


def result_line(args, n_gpus, n, n_launch, workload, rest, dev, reinit_ms, scaling, extra_config):
    value = n * args.steps / dev["elapsed"]
    pairs = (dev["hit_pairs"] + n) / n
    pair, rb = rooflines(n_launch, n, dev, pairs)
    frame_ms = 10 * dev["ms_per_step"] + reinit_ms
    cfg = {
        "workload": workload,
        "particles": n, "sweeps_per_step": SWEEPS_PER_STEP, "force_evals_per_s": value * SWEEPS_PER_STEP,
        "flow": "developed: %d steps advanced untimed before the warm-up (t = %.3f s at the end of the timed region)"
                % (args.preroll, dev["t"]),
        "pairs_in_radius_per_particle": pairs,
        "list_rebuilds": dev["rebuilds"], "rebuild_ms_each": dev["rebuild_ms_each"],
        "step_list_draws": dev["draws"], "draw_ms_each": dev["draw_ms_each"],
        "pair_launches_on_step_lists": dev["launches_on_step_lists"],
        "step_redos": dev["step_redos"],
        "at_rest": {
            "value": n * rest["steps"] / rest["elapsed"], "steps": rest["steps"], "ms_per_step": rest["ms_per_step"],
            "pair_kernel_ms": rest["pair_ms"], "list_rebuilds": rest["rebuilds"],
            "roofline_frac": pair_bytes_per_particle(n_launch == n) * n_launch / (rest["pair_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "the lattice at rest from t = 0: no list rebuild falls into the region (best case)",
        },
        "reinit_ms": reinit_ms,
        "frame": {"what": "10 steps + 1 DensityReinit, the reference's frame loop (WCXSPH3D.cpp:475-482)",
                  "ms": frame_ms, "particle_steps_per_s": n * 10 / (frame_ms * 1e-3)},
    }
    cfg.update(extra_config)
    out = {
        "metric": "particle-steps/sec (3-D WCXSPH dam break)",
        "value": value, "unit": "particle-steps/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dev["ms_per_step"],
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": cfg, "roofline": pair,
    }
    if rb:
        out["roofline_rebuild"] = rb
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preroll", type=int, default=600,
                    help="steps advanced untimed before the warm-up, so that the timed region is developed flow")
    ap.add_argument("--fluid", type=str, default="",
                    help="fluid block NXxNYxNZ: per GPU (weak scaling, default 640x320x48) or in all "
                         "(strong scaling, default 2048x1024x48)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="several GPUs: grow the case with the ranks (weak) or cut one fixed case into slabs (strong)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-strong", action="store_true",
                    help="several GPUs, weak scaling: skip the strong-scaling leg reported beside the headline")
    ap.add_argument("--strong-fluid", type=str, default="",
                    help="fluid block of the strong-scaling leg (default 2048x1024x48, the >=100 M-particle case)")
    ap.add_argument("--strong-steps", type=int, default=10, help="timed steps of the strong-scaling leg")
    ap.add_argument("--strong-child", action="store_true",
                    help="internal: this process is one rank of the strong-scaling leg, started by a rank of the run proper")
    ap.add_argument("--skin", type=float, default=0.0, help="tuning: candidate-list skin fraction (0 = follow the motion)")
    ap.add_argument("--inner", type=float, default=0.0, help="tuning: step-list margin (0 = automatic, <0 = no step lists)")
    args = ap.parse_args()

    import torch
    from sph_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pair path has no CPU fallback")
    # one GPU per rank.  Ranks may share a GPU only over the host-staged transport (a development
    # box): RCCL refuses two ranks on one device, so say so before anything touches a GPU.
    ndev = torch.cuda.device_count()
    if world > ndev and os.environ.get("SPH_TRANSPORT", "rccl") != "host":
        raise SystemExit("WORLD_SIZE=%d but %d GPU(s) visible: one rank per GPU (or SPH_TRANSPORT=host to let ranks "
                         "share a GPU over host-staged buffers)" % (world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)

    if not args.fluid:
        args.fluid = "%dx%dx%d" % (C4 if (world > 1 and args.scaling == "strong") else C3)
    nx, ny, nz = (int(v) for v in args.fluid.lower().split("x"))
    if world > 1:
        from sph_amd import domain
        if args.strong_child:
            return domain.strong_child_main(args, world, rank, local_rank)
        return domain.bench_main(args, world, rank, local_rank, (nx, ny, nz))

    case = dam_break_3d(nx, ny, nz)
    t_setup = time.perf_counter()
    sim = capi.Sim.from_case(case, device=local_rank, skin_frac=args.skin, inner_skin_frac=args.inner)
    sim.forces(capi.STATE_N)            # main(): Forces at t=0, wall f zeroed
    sim.zero_boundary_forces(capi.STATE_N)
    n = sim.npts
    setup_s = time.perf_counter() - t_setup

    rest, dev, reinit_ms = Legs(sim, args, torch.cuda.synchronize, lambda v: v).run()
    name = {C3: "C3", C4: "C4 on one GPU"}.get((nx, ny, nz), "custom size")
    out = result_line(args, 1, n, n, "%s: WCXSPH3D dam break, Pstep 0.1, fluid %dx%dx%d + walls, box %.1fx%.1fx%.1f, "
                      "developed flow" % (name, nx, ny, nz, case.box[0], case.box[1], case.box[2]),
                      rest, dev, reinit_ms, "weak", {"wall_particles": sim.bound_parts, "setup_s": setup_s})
    out["roofline"]["traffic"] = measured_traffic((nx, ny, nz), n)
    out["roofline"]["traffic_unit"] = ("bytes per launch (rocprofv3 FETCH_SIZE/WRITE_SIZE of this workload and this kernel "
                                       "source, profiles/pair_traffic.json; null when the committed counter run is of another)")
    out["roofline"]["issue"] = measured_issue((nx, ny, nz), n)
    sim.close()
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
