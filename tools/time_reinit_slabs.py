"""Development check: DensityReinit on a slab-decomposed run (ranks sharing the one GPU, bytes
staged through gloo): time per call and the number of exchange rounds.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/time_reinit_slabs.py [nx ny nz]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import bench
from sph_amd import capi, domain

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (160, 160, 48)
case = bench.dam_break_3d(nx * world, ny, nz)
sim = domain.SlabSim.from_case(case, rank, world, device=0)
tr = domain.HostTransport(rank, world)
sim.set_transport(tr.struct)
sim.forces(capi.STATE_N)
sim.zero_boundary_forces(capi.STATE_N)
sim.step(2)
for k in range(2):
    dist.barrier()
    e0, t0 = tr.exchanges, time.perf_counter()
    sim.density_reinit()
    dt = time.perf_counter() - t0
    if rank == 0:
        info = sim.reinit_info()
        print("%d ranks, %d particles (%d on rank 0): DensityReinit %.1f ms, %d exchanges; walk %d, %d launches, %d rounds%s"
              % (world, sim.npts_global, sim.npts, dt * 1e3, tr.exchanges - e0, info["walk"], info["launches"], info["rounds"],
                 ", ranks in turn" if info["took_turns"] else ""), flush=True)
    sim.step(1)
sim.close()
dist.barrier()
dist.destroy_process_group()
