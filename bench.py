"""Headline benchmark: particle-steps/s of the 3-D WCXSPH dam break.

One "step" is one full Newmark-beta call of the reference (WCXSPH3D.cpp:251-303): eight
sweeps, each a neighbour search + Forces + update.  Inputs are resident in HBM before the
timed region.  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# MI355X peaks used for the roofline fractions
HBM_PEAK_GBS = 8000.0      # HBM3E spec
FP64_PEAK_TFLOPS = 78.6    # FP64 vector spec
# Algorithmic traffic of the fused pair+update kernel per particle and sweep, 3-D, reference
# precisions: pair stage 112 B (read x,v,rho,p,m 60 B; write f,V,Rrho 52 B) + integrate/EOS
# stage 212 B.  Algorithmic flops: 80 per pair inside the radius (self included).
B_ALG_PAIR = 112.0
B_ALG_UPDATE = 212.0
FLOP_PER_PAIR = 80.0


def dam_break_3d(nx, ny, nz):
    """3-D WCXSPH3D physics, Pstep 0.1 kept (H does not scale with Pstep in the 3-D program),
    fluid block nx*ny*nz in a wide shallow box twice as long as the block."""
    from sph_amd import capi
    pstep = 0.1
    start = (0.15, 0.15, 0.15)
    box = (2 * nx * pstep + 0.3, ny * pstep + 0.3, 2 * nz * pstep)
    return capi.make_case(capi.PROGRAM_WCXSPH3D, (nx, ny, nz), pstep, box, start)


def measured_traffic(fluid, n):
    """HBM bytes per launch of the pair kernel from the committed counter run of this same
    workload (rocprofv3 cannot run inside this process): FETCH_SIZE with its gfx950 correction
    + WRITE_SIZE.  None when no counter run matches the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pair_traffic.json")) as fh:
            t = json.load(fh)
    except OSError:
        return None
    if list(t["workload"]["fluid"]) != list(fluid) or t["particles"] != n:
        return None
    reads = t["fetch_size_kib"] * 1024.0 + 0.5 * t["streamed_read_bytes_per_particle"] * n
    return reads + t["write_size_kib"] * 1024.0


def cpu_baseline(sample=(64, 48, 32), steps=4):
    """The oracle (a port: the reference itself needs Eigen+nanoflann) on one host core, on
    a reduced block of the same case."""
    from oracle import oracle as O
    from sph_amd import capi
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
        "sample": "CPU restatement of jammy4536/SPH (original needs Eigen+nanoflann): "
                  "3-D dam break, fluid %dx%dx%d, %d particles, %d step(s), %.1f s"
                  % (sample[0], sample[1], sample[2], n, steps, dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--fluid", type=str, default="640x320x48", help="fluid block per GPU, NXxNYxNZ")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--skin", type=float, default=0.0, help="tuning: candidate-list skin fraction (0 = library default)")
    ap.add_argument("--inner", type=float, default=0.0, help="tuning: working-list skin fraction (0 = default, <0 = off)")
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
    # one GPU per rank; ranks wrap round only on a development box with fewer GPUs than ranks
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    nx, ny, nz = (int(v) for v in args.fluid.lower().split("x"))
    if world > 1:
        from sph_amd import domain
        return domain.bench_main(args, world, rank, local_rank, (nx, ny, nz))

    case = dam_break_3d(nx, ny, nz)
    t_setup = time.perf_counter()
    sim = capi.Sim.from_case(case, device=local_rank, skin_frac=args.skin, inner_skin_frac=args.inner)
    sim.forces(capi.STATE_N)            # main(): Forces at t=0, wall f zeroed
    sim.zero_boundary_forces(capi.STATE_N)
    n = sim.npts
    setup_s = time.perf_counter() - t_setup

    for _ in range(args.warmup):
        sim.step(1)
    sim.profile_enable(True)
    sim.profile_read()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc = sim.step(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = sim.profile_read()
    sim.profile_enable(False)

    value = n * args.steps / elapsed
    pair_ms = prof["pair_ms"] / max(prof["pair_launches"], 1)
    b_alg = (B_ALG_PAIR + B_ALG_UPDATE) * n
    f_alg = FLOP_PER_PAIR * (sc["hit_pairs"] + n)
    ach_gbs = b_alg / (pair_ms * 1e-3) / 1e9
    ach_tf = f_alg / (pair_ms * 1e-3) / 1e12
    out = {
        "metric": "particle-steps/sec (3-D WCXSPH dam break)",
        "value": value, "unit": "particle-steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "%s: WCXSPH3D dam break, Pstep 0.1, fluid %dx%dx%d + walls, box %.1fx%.1fx%.1f"
                        % ({(640, 320, 48): "C3", (2048, 1024, 48): "C4 on one GPU"}.get((nx, ny, nz), "custom size"),
                           nx, ny, nz, case.box[0], case.box[1], case.box[2]),
            "particles": n, "wall_particles": sim.bound_parts, "sweeps_per_step": 8,
            "force_evals_per_s": value * 8,
            "pairs_in_radius_per_particle": (sc["hit_pairs"] + n) / n,
            "list_rebuilds": prof["rebuilds"], "rebuild_ms_each": prof["rebuild_ms"] / max(prof["rebuilds"], 1),
            "step_redos": sc["step_redos"], "setup_s": setup_s,
        },
        "roofline": {
            "kernel": "k_pair<3,xsph,fused> (Forces + wall reset + Newmark update + EOS, one launch per sweep)",
            "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach_gbs / HBM_PEAK_GBS, "traffic": measured_traffic((nx, ny, nz), n),
            "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE/WRITE_SIZE of this workload, profiles/pair_traffic.json)",
            "algorithmic_bytes_per_launch": b_alg,
            "kernel_ms": pair_ms, "launches": prof["pair_launches"],
            "algorithmic_bytes_per_particle_sweep": B_ALG_PAIR + B_ALG_UPDATE,
            "fp64": {"achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach_tf / FP64_PEAK_TFLOPS, "flop_per_pair": FLOP_PER_PAIR},
        },
    }
    sim.close()
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
