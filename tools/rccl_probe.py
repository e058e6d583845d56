"""Probe: does the library's RCCL transport come up with two ranks that share one GPU?
(Development aid for boxes with a single GPU; real runs use one GPU per rank.)

  RANK=r WORLD_SIZE=2 MASTER_PORT=p python tools/rccl_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from sph_amd import capi, domain
    import util

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, case = util.both_cases(util.small_cases()["wcxsph3d_block"])
    sim = domain.SlabSim.from_case(case, rank, world, device=0)
    ident = [capi.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0)
    print("rank %d: id ok, installing" % rank, flush=True)
    sim.use_rccl(ident[0])
    print("rank %d: communicator up" % rank, flush=True)
    sim.forces(capi.STATE_N)
    sim.zero_boundary_forces(capi.STATE_N)
    sc = sim.step(2)
    print("rank %d: stepped, dt %.17g maxf %.17g" % (rank, sc["dt"], sc["maxf"]), flush=True)
    sim.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
