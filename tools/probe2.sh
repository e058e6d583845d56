#!/bin/bash
# two ranks of tools/rccl_probe.py on one GPU
PORT=29731
NCCL_DEBUG=WARN RANK=0 WORLD_SIZE=2 MASTER_PORT=$PORT timeout -k 5 90 python -u tools/rccl_probe.py > gpurun_out/probe0.log 2>&1 &
P0=$!
RANK=1 WORLD_SIZE=2 MASTER_PORT=$PORT timeout -k 5 90 python -u tools/rccl_probe.py > gpurun_out/probe1.log 2>&1 &
P1=$!
wait $P0; echo "rank0 rc=$?"
wait $P1; echo "rank1 rc=$?"
tail -15 gpurun_out/probe0.log gpurun_out/probe1.log
