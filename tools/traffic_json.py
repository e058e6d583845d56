"""pair_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh: means per
dispatch of the pair kernel over the last 40 dispatches of the default bench run (developed flow),
tagged with the hash of the kernel source they were measured on."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

root = sys.argv[1]
mean = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(root, "pair_" + name, "*", "*_counter_collection.csv")):
        with open(f) as fh:
            rows += [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(fh)
                     if "k_pair" in r["Kernel_Name"] and r["Counter_Name"] == name]
    rows.sort()
    last = [v for _, v in rows[-40:]]
    mean[name] = sum(last) / len(last)
with open(os.path.join(root, "bench_default.json")) as fh:
    line = json.load(fh)
print(json.dumps({
    "source": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `python bench.py --steps 5 --warmup 2 "
              "--no-cpu`, mean of the last 40 dispatches of k_pair (developed flow, sweeps through the step lists)",
    "kernel_source_sha256_16": bench.kernel_source_hash(),
    "workload": {"fluid": list(bench.C3), "n_gpus": 1},
    "particles": line["config"]["particles"],
    "fetch_size_kib": mean["FETCH_SIZE"],
    "write_size_kib": mean["WRITE_SIZE"],
    # what a particle's quad reads as contiguous streams: its own 64-byte record, count and id,
    # and (fused update) state n: f, v, V, x and the rho/Rrho pairs
    "streamed_read_bytes_per_particle": 200,
    "note": "FETCH_SIZE counts coalesced streams at one half and scattered 64-byte reads (neighbour records, list rows) "
            "in full (calibrated with tools/fetch_calib.hip)",
}, indent=1))
