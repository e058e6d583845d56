"""CPU tests of the slab decomposition's host side: plane choice, per-rank case set-up, and the
host-buffer transport over a world_size-2 (and 3) gloo group, driven through the same C
function pointers the library calls."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from sph_amd import capi, domain

import util


def test_slab_pieces_add_up_to_the_case():
    for name, spec in util.small_cases().items():
        _, pc = util.both_cases(spec)
        p, whole = capi.case_init(pc)
        for nranks in (1, 2, 3):
            planes = domain.choose_planes(pc, nranks)
            assert len(planes) == nranks + 1 and planes[0] == -np.inf and planes[-1] == np.inf
            seen = np.zeros(int(p.npts), np.int32)
            for r in range(nranks):
                st, gids = capi.case_init_range(pc, planes[r], planes[r + 1])
                assert len(gids) > 0, (name, nranks, r)
                assert np.all(np.diff(gids.astype(np.int64)) > 0)  # InitSPH order kept
                seen[gids] += 1
                assert np.array_equal(st["x"], whole["x"][:, gids])
                assert np.array_equal(st["m"], whole["m"][gids])
                x = st["x"][0]
                assert np.all(x >= planes[r]) and np.all(x < planes[r + 1])
            assert np.all(seen == 1), (name, nranks)


def test_planes_balance_the_weighted_load():
    _, pc = util.both_cases(util.small_cases()["wcxsph3d_block"])
    p, whole = capi.case_init(pc)
    wall = np.arange(int(p.npts)) < int(p.bound_parts)
    for nranks in (2, 3, 4):
        planes = domain.choose_planes(pc, nranks, wall_weight=0.4)
        loads = []
        for r in range(nranks):
            _, g = capi.case_init_range(pc, planes[r], planes[r + 1])
            loads.append(np.sum(np.where(wall[g], 0.4, 1.0)))
        # lattice layers are indivisible: allow one fluid layer (10x8 particles) of imbalance either way
        assert max(loads) - min(loads) <= 2 * 80 + 0.4 * 400, (nranks, loads)


def test_histogram_counts_every_particle():
    _, pc = util.both_cases(util.small_cases()["wcxsph_fine"])
    p, _, _ = capi.case_params(pc)
    fl, wl = capi.case_histogram_x(pc, 257)
    assert int(fl.sum()) == int(p.npts) - int(p.bound_parts)
    assert int(wl.sum()) == int(p.bound_parts)


# ---- transport over gloo, world_size > 1 -------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        tr = domain.HostTransport(rank, world)
        t = tr.struct
        # records of 64 bytes, a different count to each side, as the halo exchange sends them
        nL = 64 * (3 + rank) if rank > 0 else 0            # to the left neighbour
        nR = 64 * (5 + rank) if rank < world - 1 else 0    # to the right neighbour
        rL = 64 * (5 + rank - 1) if rank > 0 else 0        # what the left neighbour sends right
        rR = 64 * (3 + rank + 1) if rank < world - 1 else 0
        sL = np.full(max(nL, 1), 10 * rank + 1, np.uint8)
        sR = np.full(max(nR, 1), 10 * rank + 2, np.uint8)
        bL = np.zeros(max(rL, 1), np.uint8)
        bR = np.zeros(max(rR, 1), np.uint8)
        rc = t.sendrecv(None, sL.ctypes.data, nL, bL.ctypes.data, rL, sR.ctypes.data, nR, bR.ctypes.data, rR, None)
        assert rc == 0, tr.error
        if rank > 0:
            assert np.all(bL[:rL] == 10 * (rank - 1) + 2)
        if rank < world - 1:
            assert np.all(bR[:rR] == 10 * (rank + 1) + 1)
        v = (C.c_double * 3)(float(rank), -float(rank), 1.5)
        assert t.allreduce(None, v, 3, 0, None) == 0
        assert list(v) == [world - 1.0, 0.0, 1.5]
        v = (C.c_double * 2)(float(rank + 1), 0.25)
        assert t.allreduce(None, v, 2, 1, None) == 0
        assert list(v) == [world * (world + 1) / 2.0, 0.25 * world]
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover - reported to the parent
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_host_transport_over_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in results), results


# ---- the bench's strong-scaling leg: processes of its own, watched by the ranks together --------

_CHILD = """
import os, sys, time
rank, how = int(os.environ["RANK"]), sys.argv[1]
assert "TORCHELASTIC_RUN_ID" not in os.environ and os.environ["SPH_TRANSPORT"] == "host"
if how == "fine":
    time.sleep(0.3 * rank)
    if rank == 0:
        print("noise on stdout")
        print('{"value": 42.0, "port": %s}' % os.environ["MASTER_PORT"])
elif how == "one fails":
    if rank == 1:
        sys.exit(3)
    time.sleep(120)  # waits for the rank that is gone: must be stopped long before
else:
    time.sleep(120)
"""


def _leg_main(rank, world, port, how, q):
    try:
        import sys
        import time
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          TORCHELASTIC_RUN_ID="the launcher's", SPH_STRONG_LIMIT_S="2")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t0 = time.perf_counter()
        res = domain.strong_leg_in_children(None, world, rank, "host", dist, cmd=[sys.executable, "-c", _CHILD, how])
        took = time.perf_counter() - t0
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, res, took))
    except Exception as e:  # pragma: no cover - reported to the parent
        import traceback
        q.put((rank, traceback.format_exc() + repr(e), 0.0))


@pytest.mark.parametrize("how", ["fine", "one fails", "all hang"])
def test_strong_leg_processes_are_watched_and_stopped_together(how):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_leg_main, args=(r, 2, port, how, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    (_, res0, took0), (_, res1, took1) = results
    assert max(took0, took1) < 30  # nobody sat out a process that waited for a dead one
    if how == "fine":
        assert res0["value"] == 42.0 and res0["port"] != port and res1 is None
    else:
        assert isinstance(res0, dict) and res0 == res1 and "error" in res0, results
        assert ("ended with code 3" in res0["error"]) == (how == "one fails")
        assert ("did not finish" in res0["error"]) == (how == "all hang")
