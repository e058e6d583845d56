"""Slab decomposition along x: one process and one context per GPU.

The reference is one process on one core and has nothing like this.  Each rank owns the
particles whose x lay in its interval at set-up; the library fetches the neighbouring slabs'
boundary particles (the halo) before every Forces evaluation, and the ranks agree once per
step on max(mu), max|f|, the convergence sums and on whether the step must be redone.

Two transports move the bytes:
  * the library's own RCCL transport (`Sim.use_rccl`), device buffers, for real multi-GPU runs;
  * `HostTransport` below, host buffers over a torch.distributed process group (gloo), used
    by the tests and wherever ranks share one GPU.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

from . import capi


# ---------------------------------------------------------------------------------------------
# choosing the planes


def choose_planes(case, nranks, wall_weight=0.4, nbins=16384):
    """x-planes p[0] < ... < p[nranks] with p[0] = -inf, p[nranks] = +inf such that the slabs
    carry about equal work.  A wall particle costs `wall_weight` of a fluid particle (pairs of
    two resting wall particles are never evaluated)."""
    if nranks == 1:
        return [-np.inf, np.inf]
    fl, wl = capi.case_histogram_x(case, nbins)
    w = fl.astype(np.float64) + wall_weight * wl.astype(np.float64)
    cum = np.cumsum(w)
    edges = np.linspace(0.0, case.box[0], nbins + 1)
    planes = [-np.inf]
    for r in range(1, nranks):
        target = cum[-1] * r / nranks
        b = int(np.searchsorted(cum, target))
        planes.append(float(edges[min(b + 1, nbins)]))
    planes.append(np.inf)
    for a, b in zip(planes[:-1], planes[1:]):
        if not b > a:
            raise capi.SphError("cannot cut %d slabs out of this case" % nranks)
    return planes


# ---------------------------------------------------------------------------------------------
# host-buffer transport over a torch.distributed group


def _view(ptr, nbytes):
    import torch
    buf = (C.c_ubyte * nbytes).from_address(ptr)
    return torch.frombuffer(buf, dtype=torch.uint8)


class HostTransport:
    """sendrecv / allreduce callbacks on host memory through torch.distributed (any backend
    that takes CPU tensors; gloo in practice)."""

    def __init__(self, rank, nranks, group=None):
        import torch.distributed as dist
        self.dist, self.rank, self.nranks, self.group = dist, rank, nranks, group
        self.error = None
        self.exchanges = 0  # sendrecv calls so far
        self._sr = capi.SENDRECV_FN(self._sendrecv)
        self._ar = capi.ALLREDUCE_FN(self._allreduce)
        self.struct = capi.make_transport(1, self._sr, self._ar)

    def _sendrecv(self, user, sL, nsL, rL, nrL, sR, nsR, rR, nrR, stream):
        try:
            self.exchanges += 1
            d, ops = self.dist, []
            if nsL:
                ops.append(d.P2POp(d.isend, _view(sL, nsL), self.rank - 1, self.group))
            if nsR:
                ops.append(d.P2POp(d.isend, _view(sR, nsR), self.rank + 1, self.group))
            if nrL:
                ops.append(d.P2POp(d.irecv, _view(rL, nrL), self.rank - 1, self.group))
            if nrR:
                ops.append(d.P2POp(d.irecv, _view(rR, nrR), self.rank + 1, self.group))
            if ops:
                for w in d.batch_isend_irecv(ops):
                    w.wait()
            return 0
        except Exception as e:  # a callback must not raise into C
            self.error = e
            return 1

    def _allreduce(self, user, vals, n, op, stream):
        try:
            import torch
            d = self.dist
            buf = (C.c_double * n).from_address(C.addressof(vals.contents))
            t = torch.frombuffer(buf, dtype=torch.float64)
            d.all_reduce(t, op=d.ReduceOp.SUM if op else d.ReduceOp.MAX, group=self.group)
            return 0
        except Exception as e:
            self.error = e
            return 1


class DeviceTransport(HostTransport):
    """The device-buffer form of the callbacks (host_buffers = 0, as the RCCL transport uses it),
    moved by hand: wait for the stream, copy to the host, exchange over the process group, copy
    back.  For ranks that share a GPU, where RCCL refuses to form a communicator."""

    def __init__(self, rank, nranks, group=None):
        super().__init__(rank, nranks, group)
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._sr_dev = capi.SENDRECV_FN(self._sendrecv_dev)
        self._ar_dev = capi.ALLREDUCE_FN(self._allreduce_dev)
        self.struct = capi.make_transport(0, self._sr_dev, self._ar_dev)

    def _to_host(self, dptr, n):
        buf = (C.c_ubyte * max(n, 1))()
        if n and self.hip.hipMemcpy(buf, dptr, n, 2) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        return buf

    def _sendrecv_dev(self, user, sL, nsL, rL, nrL, sR, nsR, rR, nrR, stream):
        try:
            if self.hip.hipStreamSynchronize(stream) != 0:
                raise RuntimeError("hipStreamSynchronize failed")
            hsL, hsR = self._to_host(sL, nsL), self._to_host(sR, nsR)
            hrL, hrR = (C.c_ubyte * max(nrL, 1))(), (C.c_ubyte * max(nrR, 1))()
            rc = self._sendrecv(user, C.addressof(hsL), nsL, C.addressof(hrL), nrL, C.addressof(hsR), nsR,
                                C.addressof(hrR), nrR, None)
            if rc:
                return rc
            for dptr, hbuf, n in ((rL, hrL, nrL), (rR, hrR, nrR)):
                if n and self.hip.hipMemcpy(dptr, hbuf, n, 1) != 0:
                    raise RuntimeError("hipMemcpy H2D failed")
            return 0
        except Exception as e:
            self.error = e
            return 1

    def _allreduce_dev(self, user, vals, n, op, stream):
        try:
            if self.hip.hipStreamSynchronize(stream) != 0:
                raise RuntimeError("hipStreamSynchronize failed")
            dptr = C.cast(vals, C.c_void_p)
            h = (C.c_double * n)()
            if self.hip.hipMemcpy(h, dptr, 8 * n, 2) != 0:
                raise RuntimeError("hipMemcpy D2H failed")
            rc = self._allreduce(user, C.cast(h, C.POINTER(C.c_double)), n, op, None)
            if rc:
                return rc
            if self.hip.hipMemcpy(dptr, h, 8 * n, 1) != 0:
                raise RuntimeError("hipMemcpy H2D failed")
            return 0
        except Exception as e:
            self.error = e
            return 1


# ---------------------------------------------------------------------------------------------


class SlabSim(capi.Sim):
    """This rank's slab of a case.  Every method that evaluates forces, steps or probes the
    neighbour sets is collective: all ranks call it, in the same order."""

    @classmethod
    def from_case(cls, case, rank, nranks, planes=None, device=-1, halo_room=None, migrate=True, **overrides):
        planes = planes or choose_planes(case, nranks)
        p, _, _ = capi.case_params(case)
        st, gids = capi.case_init_range(case, planes[rank], planes[rank + 1])
        n = len(gids)
        if n == 0:
            raise capi.SphError("rank %d owns no particle" % rank)
        p.npts_global = p.npts
        p.npts = n
        p.rank, p.nranks = rank, nranks
        # a rank of up to 32 M particles gets room for half as many again to migrate in (a collapsing
        # column fills the slabs ahead of it); a larger one a quarter, which still fits the GPU
        room = halo_room if halo_room is not None else max(1 << 16, n // 4 if n <= (32 << 20) else n // 8)
        p.capacity = n + 4 * room  # growth by migration 2*room, each side's halo room
        p.migrate = 1 if migrate else 0
        big = 1e300
        p.slab_lo = max(planes[rank], -big)
        p.slab_hi = min(planes[rank + 1], big)
        p.device = device
        for k, v in overrides.items():
            setattr(p, k, v)
        sim = cls(p)
        sim._npts0 = n
        sim.global_ids = gids
        sim.npts_global = int(p.npts_global)
        sim.upload(capi.STATE_N, st)
        sim.upload(capi.STATE_NP1, st)
        sim.upload_global_ids(gids)
        return sim

    def refresh(self):
        """After steps with migration on: the owned set (and with it the order and length of
        the upload/download arrays) may have changed."""
        n, self.migrated_in, self.migrated_out = self.local_count()
        if n != self.npts or self.migrated_in or self.migrated_out:
            self.npts = n
            self.global_ids = self.download_global_ids()

    def step(self, nsteps=1):
        sc = super().step(nsteps)
        self.refresh()
        return sc

    def rebalance(self, dist, group=None, wall_weight=0.4, nbins=4096, max_fraction=0.5):
        """Collective: move the planes towards equal weighted particle counts, judged on where
        the particles are now.  A plane moves at most as far as `max_fraction` of the per-side
        hand-over room allows (particles change owner one slab per list rebuild), so a large
        imbalance is worked off over several calls.  Needs migrate on.  Returns the planes."""
        x = self.download(capi.STATE_NP1, names=("x",))["x"][0]
        wall = self.global_ids < self.bound_parts
        lohi = [None] * self.params.nranks
        dist.all_gather_object(lohi, (float(x.min()), float(x.max()), float(self.params.slab_lo),
                                      float(self.params.slab_hi)), group=group)
        lo = min(v[0] for v in lohi)
        hi = max(v[1] for v in lohi)
        edges = np.linspace(lo, hi + 1e-9 * max(1.0, abs(hi)), nbins + 1)
        h = np.histogram(x, bins=edges, weights=np.where(wall, wall_weight, 1.0))[0]
        cnt = np.histogram(x, bins=edges)[0].astype(np.float64)
        both = [None] * self.params.nranks
        dist.all_gather_object(both, (h, cnt), group=group)
        h = sum(b[0] for b in both)
        cnt = sum(b[1] for b in both)
        cum, ccum = np.cumsum(h), np.cumsum(cnt)
        nranks, rank = self.params.nranks, self.params.rank
        room = max_fraction * (int(self.params.capacity) - self._npts0) / 4.0
        planes = [lohi[0][2]]
        for r in range(1, nranks):
            old = lohi[r][2]
            want = float(edges[min(int(np.searchsorted(cum, cum[-1] * r / nranks)) + 1, nbins)])
            # particles between the old and the wanted plane may not exceed the room: walk back
            def between(a, b):
                ia, ib = sorted((int(np.searchsorted(edges, a)), int(np.searchsorted(edges, b))))
                return ccum[min(ib, nbins) - 1] - (ccum[ia - 1] if ia > 0 else 0.0) if ib > ia else 0.0
            new = want
            for _ in range(40):
                if between(old, new) <= room:
                    break
                new = 0.5 * (old + new)
            else:
                new = old
            planes.append(new)
        planes.append(lohi[-1][3])
        for a, b in zip(planes[:-1], planes[1:]):
            if not b > a:
                return [v[2] for v in lohi] + [lohi[-1][3]]  # no change rather than an empty slab
        self.set_slab_interval(planes[rank], planes[rank + 1])
        self.params.slab_lo, self.params.slab_hi = planes[rank], planes[rank + 1]
        return planes


def gather_state(sim, state, dist, group=None):
    """All ranks' downloaded `state` dicts merged into global original order (on every rank)."""
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, (sim.global_ids, state), group=group)
    n = sim.npts_global
    out = {}
    for k, a in state.items():
        shape = (a.shape[0], n) if a.ndim == 2 else (n,)
        out[k] = np.zeros(shape, a.dtype)
    for g, st in parts:
        for k, a in st.items():
            out[k][..., g] = a
    return out


# ---------------------------------------------------------------------------------------------
# multi-GPU benchmark: weak scaling (the fluid block and the box grow along x with the ranks) or
# strong scaling (one fixed case cut into as many slabs as there are ranks)


def bench_main(args, world, rank, local_rank, fluid):
    import threading

    import torch
    import torch.distributed as dist
    import bench as B

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # a transport that never returns must not hold the node for ever: give up loudly
    limit_s = int(os.environ.get("SPH_BENCH_LIMIT_S", "1500"))

    def give_up():
        print("rank %d: bench did not finish within %d s, giving up" % (rank, limit_s), file=sys.stderr, flush=True)
        os._exit(3)

    watchdog = threading.Timer(limit_s, give_up)
    watchdog.daemon = True
    watchdog.start()
    # control plane (id broadcast, barriers, timing) on gloo; particle data moves over RCCL
    # gloo reports its connections on stdout, which is reserved for the one result line: send what
    # is written there while the group forms (and makes its first connections) to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    nx, ny, nz = fluid
    strong = args.scaling == "strong"
    case = B.dam_break_3d(nx if strong else nx * world, ny, nz)
    t_setup = time.perf_counter()

    def all_ok(err):
        """A rank that failed must not leave the others waiting in a collective: agree first."""
        flag = torch.tensor([1 if err else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if err:
            print("rank %d: %s" % (rank, err), file=sys.stderr, flush=True)
        if int(flag[0]):
            dist.destroy_process_group()
            raise SystemExit(1)

    sim, err = None, None
    try:
        sim = SlabSim.from_case(case, rank, world, device=local_rank, skin_frac=args.skin, inner_skin_frac=args.inner)
    except Exception as e:  # reported below, on every rank
        err = repr(e)
    all_ok(err)
    asked = os.environ.get("SPH_TRANSPORT", "rccl")
    if asked not in ("rccl", "host"):
        raise SystemExit("SPH_TRANSPORT must be rccl or host")
    transport, why_not_rccl, rccl_thread = asked, None, None
    if asked == "rccl":
        ident = [capi.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        # creating the communicator is a collective inside RCCL: if it does not return, every
        # rank gives up after the same time
        got = {}

        def create():
            try:
                sim.use_rccl(ident[0])
            except capi.SphError as e:
                got["why"] = str(e)

        wait_s = float(os.environ.get("SPH_RCCL_INIT_S", "120"))
        rccl_thread = threading.Thread(target=create, daemon=True)
        rccl_thread.start()
        rccl_thread.join(wait_s)
        why = got.get("why")
        if rccl_thread.is_alive():
            why = "ncclCommInitRank did not return within %.0f s" % wait_s
        failed = torch.tensor([1 if why else 0], dtype=torch.int32)
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if int(failed[0]):
            # no communicator (e.g. ranks sharing a GPU).  The run goes on over the staged transport
            # so that there is a line to read, says so in `transport`, and ends with a failure code.
            why_not_rccl = why or "another rank could not create the communicator"
            if rank == 0:
                print("RCCL transport unavailable (%s); halo bytes staged through the host, exit code 4"
                      % why_not_rccl, file=sys.stderr, flush=True)
            transport = "host"
    if transport == "host":  # ranks sharing a GPU on a development box: staged through gloo
        host_tr = HostTransport(rank, world)
        sim.set_transport(host_tr.struct)
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    setup_s = time.perf_counter() - t_setup
    all_ok(None)

    def sync():
        torch.cuda.synchronize()
        dist.barrier()

    def agree_max(vals):
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    rest, dev, reinit_ms = B.Legs(sim, args, sync, agree_max).run()
    sent, recvd = sim.halo_counts()
    info = torch.tensor([sim.npts, sent[0] + sent[1]], dtype=torch.int64)
    alli = [torch.zeros_like(info) for _ in range(world)]
    dist.all_gather(alli, info)
    n = sim.npts_global

    # Beside a weak-scaling headline: the >=100 M-particle case cut into as many slabs as there are
    # ranks, a few timed steps of developed flow, against the committed single-GPU figure of the
    # same case -- the strong-scaling efficiency, in the same line.
    strong_leg = None
    if not strong and not args.no_strong:
        sim.close()
        sim = None
        strong_leg = strong_leg_in_children(args, world, rank, transport, dist)

    if rank == 0:
        per = [int(c[0]) for c in alli]
        halo = [int(c[1]) for c in alli]
        how = "RCCL send/recv" if transport == "rccl" else "host staging + gloo"
        if strong:
            workload = ("C4 cut into %d x-slabs (strong scaling)" if fluid == B.C4 else "custom size cut into %d x-slabs (strong scaling)") % world
        else:
            workload = "C3 x %d along x (weak scaling)" % world
        workload += (": WCXSPH3D dam break, Pstep 0.1, fluid %dx%dx%d + walls, box %.1fx%.1fx%.1f, developed flow, halo exchange over %s"
                     % (case.npart[0], ny, nz, case.box[0], case.box[1], case.box[2], how))
        extra = {
            "particles_per_rank": per, "halo_particles_sent_per_rank": halo,
            "halo_bytes_per_sweep_per_rank": [64 * h for h in halo], "setup_s": setup_s,
            "exchange": dev["comm"],
        }
        if strong_leg is not None:
            extra["strong"] = strong_leg
        out = B.result_line(args, world, n, max(per), workload, rest, dev, reinit_ms, args.scaling, extra)
        out["transport"] = transport
        if why_not_rccl:
            out["transport_note"] = "RCCL was asked for and could not be created: " + why_not_rccl
        out["roofline"]["kernel"] = ("k_pair<3,xsph,fused>, the three launches of one sweep (two boundary layers + interior), "
                                     "slowest rank")
        print(json.dumps(out), flush=True)
    if rccl_thread is not None and rccl_thread.is_alive():
        # the thread is still inside ncclCommInitRank with this context: leave the context alone
        dist.barrier()
        os._exit(4)
    if sim is not None:
        sim.close()
    dist.barrier()
    dist.destroy_process_group()
    watchdog.cancel()
    if why_not_rccl:
        raise SystemExit(4)


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def strong_leg_in_children(args, world, rank, transport, dist, cmd=None):
    """The strong-scaling leg runs in processes of its own, one started by every rank (the same
    program with --strong-child, a rendezvous port of their own).  Whatever happens to them -- a
    rank out of memory or of room for migrating particles, a communicator that never comes up --
    the ranks of the run proper stay in step, wait at most SPH_STRONG_LIMIT_S seconds, and the
    headline is printed with the leg's figures or with what went wrong.  Collective."""
    import subprocess
    import bench as B
    port = [free_port() if rank == 0 else None]
    dist.broadcast_object_list(port, src=0)
    # the launcher's own rendezvous store (TORCHELASTIC_*) belongs to the run proper: the leg's ranks
    # meet at a port of their own, rank 0 of them serving it
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_PORT=str(port[0]), SPH_TRANSPORT=transport)
    if cmd is None:  # (tests of the watching give a command of their own)
        cmd = [sys.executable, os.path.abspath(B.__file__), "--strong-child", "--gpus", str(world), "--preroll", str(args.preroll),
               "--warmup", str(args.warmup), "--strong-steps", str(args.strong_steps), "--skin", str(args.skin),
               "--inner", str(args.inner), "--no-cpu"]
        if args.strong_fluid:
            cmd += ["--strong-fluid", args.strong_fluid]
    limit_s = float(os.environ.get("SPH_STRONG_LIMIT_S", "600"))
    err, text, child = None, "", None
    try:
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    except OSError as e:
        err = "rank %d could not start its process of the leg: %r" % (rank, e)
    # the ranks look at their processes together twice a second: when one of them has failed (or
    # time is up) the others' processes -- waiting for it in some exchange -- are stopped as well
    RUNNING, DONE, FAILED = 0, 1, 2
    t0 = time.perf_counter()
    while True:
        rc = child.poll() if child is not None else 1
        mine = RUNNING if rc is None else (DONE if rc == 0 and err is None else FAILED)
        if mine == FAILED and err is None:
            err = "rank %d of the leg ended with code %d" % (rank, rc)
        if mine == RUNNING and time.perf_counter() - t0 > limit_s:
            mine, err = FAILED, "rank %d of the leg did not finish within %.0f s and was stopped" % (rank, limit_s)
        states = [None] * world
        dist.all_gather_object(states, mine)
        if FAILED in states or all(v == DONE for v in states):
            break
        time.sleep(0.5)
        if rank == 0 and int(time.perf_counter() - t0) // 60 != int(time.perf_counter() - t0 - 0.5) // 60:
            print("strong-scaling leg: %.0f s so far" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
    if child is not None:
        if child.poll() is None:
            child.kill()  # this very process, nothing else
        text = child.communicate()[0].decode(errors="replace")
    errs = [None] * world
    dist.all_gather_object(errs, err)
    errs = [e for e in errs if e]
    if errs:
        return {"error": "; ".join(errs)}
    if rank != 0:
        return None
    try:
        return json.loads(text.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return {"error": "the leg printed no result line"}


def strong_child_main(args, world, rank, local_rank):
    """One rank of the strong-scaling leg (see strong_leg_in_children): rank 0 prints the leg's
    figures as one JSON line; any failure ends the process with a non-zero code."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)  # gloo reports its connections on stdout
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    def sync():
        torch.cuda.synchronize()
        dist.barrier()

    def agree_max(vals):
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    transport = os.environ.get("SPH_TRANSPORT", "rccl")
    res = strong_scaling_leg(args, world, rank, local_rank, transport, None, sync, agree_max,
                             all_ok_soft=lambda err: agree_max([1.0 if err else 0.0])[0] > 0)
    failed = "error" in res
    if rank == 0:
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if failed:
        raise SystemExit(1)


def single_gpu_anchor(fluid):
    """The committed single-GPU bench line of this fluid block (developed flow), for the strong-
    scaling efficiency: (particle-steps/s, file) or (None, None)."""
    import glob
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(here, "*_single_gpu_bench.json"))):
        try:
            with open(f) as fh:
                d = json.loads(fh.read().strip().splitlines()[-1])
        except (OSError, ValueError, IndexError):
            continue
        if d.get("n_gpus") == 1 and "%dx%dx%d" % tuple(fluid) in d.get("config", {}).get("workload", ""):
            best = (float(d["value"]), os.path.basename(f))  # the latest by name
    return best


def strong_scaling_leg(args, world, rank, local_rank, transport, ident_unused, sync, agree_max, all_ok_soft):
    """One fixed case cut into `world` slabs: pre-roll, warm-up, a few timed steps.  Collective; a
    failure on any rank is agreed on and reported in the line instead of ending the run."""
    import bench as B
    fluid = tuple(int(v) for v in args.strong_fluid.lower().split("x")) if args.strong_fluid else B.C4
    err, sim = None, None
    try:
        case = B.dam_break_3d(*fluid)
        sim = SlabSim.from_case(case, rank, world, device=local_rank, skin_frac=args.skin, inner_skin_frac=args.inner)
    except Exception as e:  # agreed on below
        err = repr(e)
    if all_ok_soft(err):
        if sim is not None:
            sim.close()
        return {"error": err or "another rank could not set the case up"}
    try:
        if transport == "rccl":
            # a communicator of its own: the first one went with the weak-scaling context
            import torch.distributed as dist
            ident = [capi.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)
            sim.use_rccl(ident[0])
        else:
            host_tr = HostTransport(rank, world)
            sim.set_transport(host_tr.struct)
            sim._strong_transport = host_tr  # keep the callbacks alive
    except Exception as e:
        err = repr(e)
    if all_ok_soft(err):
        sim.close()
        return {"error": err or "another rank could not open its transport"}
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    done = 0
    while done < args.preroll:
        m = min(20, args.preroll - done)
        sim.step(m)
        done += m
    for _ in range(args.warmup):
        sim.step(1)
    legs = B.Legs(sim, args, sync, agree_max)
    k = max(1, args.strong_steps)
    dev = legs.timed(k)
    n = sim.npts_global
    per = agree_max([float(sim.npts)])[0]
    sim.close()
    value = n * k / dev["elapsed"]
    anchor, where = single_gpu_anchor(fluid)
    return {
        "what": "fluid %dx%dx%d cut into %d x-slabs, %d steps advanced untimed, then %d timed steps (developed flow)"
                % (fluid[0], fluid[1], fluid[2], world, args.preroll, k),
        "particles": n, "most_particles_on_a_rank": int(per), "value": value, "unit": "particle-steps/s",
        "ms_per_step": dev["ms_per_step"], "pair_ms": dev["pair_ms"], "list_rebuilds": dev["rebuilds"],
        "rebuild_ms_each": dev["rebuild_ms_each"], "exchange": dev["comm"],
        "single_gpu_value": anchor, "single_gpu_source": where,
        "efficiency": (value / (world * anchor)) if anchor else None,
    }
