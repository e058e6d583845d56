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


def last40(passname, counter):
    """Mean of a counter over the last 40 dispatches of the pair kernel in one pass, or None."""
    rows = []
    for f in glob.glob(os.path.join(root, "pair_" + passname, "*", "*_counter_collection.csv")):
        with open(f) as fh:
            rows += [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(fh)
                     if "k_pair" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort()
    last = [v for _, v in rows[-40:]]
    return sum(last) / len(last) if last else None


for name in ("FETCH_SIZE", "WRITE_SIZE"):
    mean[name] = last40(name, name)
sq = {c: last40("SQ", c) for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
                                   "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS")}
ta = {c: last40("TA", c) for c in ("TA_BUSY_avr", "TA_BUSY_max", "GRBM_GUI_ACTIVE")}
issue = None
if all(v is not None for v in sq.values()) and all(v is not None for v in ta.values()):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ cycle counters count in units of four cycles;
    # 256 compute units of four SIMDs
    cycles = ta["GRBM_GUI_ACTIVE"] / 8.0
    issue = {
        "valu_busy": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles),
        "ta_busy_avg": ta["TA_BUSY_avr"] / cycles, "ta_busy_max": ta["TA_BUSY_max"] / cycles,
        "valu_instructions_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"],
        "vmem_reads_per_wave": sq["SQ_INSTS_VMEM_RD"] / sq["SQ_WAVES"],
        "lds_instructions_per_wave": sq["SQ_INSTS_LDS"] / sq["SQ_WAVES"],
        "waves": sq["SQ_WAVES"], "particles_per_wave": 64,
        "wave_share_waiting": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
        "wave_share_issue_stalled": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
        "waves_resident_per_simd": 4.0 * sq["SQ_WAVE_CYCLES"] / (1024.0 * cycles),
        "launch_cycles": cycles,
        "source": "rocprofv3 --pmc, one run per pass: SQ_* and TA_BUSY/GRBM_GUI_ACTIVE, last 40 dispatches (developed flow)",
    }
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
    # what is read per particle as contiguous streams: its own 64-byte record, count and id, and
    # (fused update) state n: f, v, V, x (96 bytes), rho and Rrho
    "streamed_read_bytes_per_particle": 176,
    "issue": issue,
    "note": "FETCH_SIZE counts coalesced streams at one half and scattered 64-byte reads (neighbour records, list rows) "
            "in full (calibrated with tools/fetch_calib.hip)",
}, indent=1))
