"""One rank of a slab-decomposed run (started by test_gpu_slabs.py, one process per rank, all
on the same GPU, bytes moved through gloo).  Rank 0 writes the merged results to --out."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", required=True, help="JSON: arguments of capi.make_case or {program}")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--skin", type=float, default=0.0)
    ap.add_argument("--jitter", type=float, default=0.0)
    ap.add_argument("--vx", type=float, default=0.0, help="initial x-velocity of the fluid")
    ap.add_argument("--migrate", type=int, default=1)
    ap.add_argument("--even-planes", type=int, default=0, help="cut the box evenly in x instead of by load")
    ap.add_argument("--rebalance", type=int, default=0, help="re-cut the planes every this many steps")
    ap.add_argument("--reinit", type=int, default=0, help="DensityReinit after every this many steps")
    ap.add_argument("--halo-room", type=int, default=0,
                    help="room for halo and migrated particles (0: 4096 when planes are re-cut, the library's choice otherwise)")
    ap.add_argument("--transport", default="host", help="host: host-buffer callbacks; device: device-buffer callbacks")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()

    import torch.distributed as dist
    from sph_amd import capi, domain

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = json.loads(args.spec)
    if "npart" in spec:
        case = capi.make_case(spec["program"], spec["npart"], spec["pstep"], spec["box"], spec["start"],
                              drop=spec.get("drop"))
    else:
        case = capi.default_case(spec["program"])
    over = {"skin_frac": args.skin} if args.skin > 0 else {}
    planes = None
    if args.even_planes:
        planes = [-np.inf] + [case.box[0] * r / world for r in range(1, world)] + [np.inf]
    sim = domain.SlabSim.from_case(case, rank, world, planes=planes, device=0, migrate=bool(args.migrate),
                                   halo_room=args.halo_room or ((1 << 12) if args.rebalance else None), **over)
    tr = domain.DeviceTransport(rank, world) if args.transport == "device" else domain.HostTransport(rank, world)
    sim.set_transport(tr.struct)
    if args.jitter > 0:
        # the same perturbation of the whole set on every rank, then each takes its own rows
        p, st = capi.case_init(case)
        rng = np.random.Generator(np.random.MT19937(4321))
        dx = rng.uniform(-args.jitter, args.jitter, size=st["x"].shape) * case.pstep
        dx[:, : int(p.bound_parts)] = 0.0
        mine = {"x": np.ascontiguousarray((st["x"] + dx)[:, sim.global_ids])}
        sim.upload(capi.STATE_N, mine, names=("x",))
        sim.upload(capi.STATE_NP1, mine, names=("x",))

    if args.vx != 0.0:
        fluid = sim.global_ids >= sim.bound_parts
        v = np.zeros((sim.dim, sim.npts))
        v[0, fluid] = args.vx
        sim.upload(capi.STATE_N, {"v": v}, names=("v",))
        sim.upload(capi.STATE_NP1, {"v": v}, names=("v",))

    res = {}
    sim.forces(capi.STATE_N)
    f0 = sim.download(capi.STATE_N, names=("f", "V", "Rrho"))
    offs, cols = sim.neighbours(capi.STATE_N)
    # gathered before stepping: migration may change who owns what
    g0 = domain.gather_state(sim, f0, dist)
    rows = [None] * world
    dist.all_gather_object(rows, (sim.global_ids, offs, cols))
    sim.zero_boundary_forces(capi.STATE_N)
    counts_before = [None] * world
    dist.all_gather_object(counts_before, sim.npts)
    if args.rebalance:
        done = 0
        while done < args.steps:
            k = min(args.rebalance, args.steps - done)
            sc = sim.step(k)
            done += k
            if done < args.steps:
                sim.rebalance(dist)
    elif args.reinit:
        # main()'s frames: steps, then DensityReinit on state n+1 (WCXSPH3D.cpp:475-482)
        done = 0
        while done < args.steps:
            k = min(args.reinit, args.steps - done)
            sc = sim.step(k)
            done += k
            sim.density_reinit()
        info = sim.reinit_info()
        res["reinit_info"] = np.array([info["walk"], info["launches"], info["rounds"], int(info["took_turns"]),
                                       int(info["standing_lists"])])
    else:
        syncs0 = sim.scalars()["host_syncs"]
        sc = sim.step(args.steps)
        res["host_syncs_in_steps"] = np.array([sc["host_syncs"] - syncs0, args.steps])
    st = sim.download(capi.STATE_N)
    if args.reinit:
        st["np1_rho"] = sim.download(capi.STATE_NP1, names=("rho",))["rho"]
    else:
        # a step ends with pn = pnp1 (WCXSPH3D.cpp:301): every field of the two states, bit for bit
        st1 = sim.download(capi.STATE_NP1)
        for k in ("x", "v", "f", "V", "rho", "p", "Rrho"):
            a, b = np.ascontiguousarray(st[k]), np.ascontiguousarray(st1[k])
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), "rank %d: state n+1 differs from state n in %s" % (rank, k)
    g1 = domain.gather_state(sim, st, dist)
    if rank == 0:
        for k, a in g0.items():
            res["forces_" + k] = a
        for k, a in g1.items():
            res["step_" + k] = a
        n = sim.npts_global
        counts = np.zeros(n, np.int64)
        for g, o, c in rows:
            counts[g] = np.diff(o.astype(np.int64))
        goffs = np.concatenate(([0], np.cumsum(counts)))
        gcols = np.zeros(int(goffs[-1]), np.uint32)
        for g, o, c in rows:
            o = o.astype(np.int64)
            for i, gi in enumerate(g):
                gcols[goffs[gi]:goffs[gi + 1]] = c[o[i]:o[i + 1]]
        res["nbr_offs"], res["nbr_cols"] = goffs, gcols
        res["scalars"] = np.array([sc["dt"], sc["t"], sc["maxmu"], sc["maxf"], sc["errsum"], sc["logbase"],
                                   float(sc["hit_pairs"]), float(sc["step_redos"]), float(sc["list_rebuilds"])])
        res["per_rank"] = np.array([len(r[0]) for r in rows])
    moved = [None] * world
    dist.all_gather_object(moved, (sim.npts, getattr(sim, "migrated_in", 0), getattr(sim, "migrated_out", 0)))
    if rank == 0:
        res["moved"] = np.array(moved)
        res["counts_before"] = np.array(counts_before)
        np.savez(args.out, **res)
    if tr.error is not None:
        raise tr.error
    sim.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d done" % rank, flush=True)


if __name__ == "__main__":
    main()
