"""bench.py keeps its contract: one JSON line from rank 0 with the agreed keys, on one GPU and
under torch.distributed.run with two ranks (sharing the GPU here, bytes staged through gloo)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check(line, n_gpus, steps, warmup, scaling="weak"):
    d = json.loads(line)
    assert KEYS <= set(d), KEYS - set(d)
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "particle-steps/s" and d["higher_is_better"] is True and d["scaling"] == scaling
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # the fraction is priced against the HBM roof; `bound` says what holds the launch (neither roof)
    assert r["roof"] == "hbm" and r["bound"] == "latency" and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "issue" in r
    rest = d["config"]["at_rest"]
    assert rest["value"] > 0 and d["config"]["reinit_ms"] > 0 and d["config"]["frame"]["ms"] > 10 * d["ms_per_step"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["particles"] * steps / (d["ms_per_step"] * steps * 1e-3)) <= 1e-6 * d["value"]
    return d


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--fluid", "48x24x12", "--steps", "2", "--warmup", "1", "--preroll", "40"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = _check(lines[0], 1, 2, 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "particle-steps/s"
    assert d["roofline"]["traffic"] is None  # counters were taken on the full-size workload only


def _two_ranks(extra, env):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "bench.py"),
                           "--gpus", "2", "--steps", "2", "--warmup", "1", "--preroll", "40"] + extra,
                          capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)


@pytest.mark.parametrize("scaling,fluid", [("weak", "48x24x12"), ("strong", "96x24x12")])
def test_two_rank_line(scaling, fluid):
    r = _two_ranks(["--fluid", fluid, "--scaling", scaling, "--strong-fluid", "96x24x12", "--strong-steps", "2"],
                   dict(os.environ, SPH_TRANSPORT="host"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = _check(lines[0], 2, 2, 1, scaling)
    assert d["transport"] == "host"
    per = d["config"]["particles_per_rank"]
    assert len(per) == 2 and sum(per) == d["config"]["particles"]
    assert all(h > 0 for h in d["config"]["halo_bytes_per_sweep_per_rank"])
    # both modes ran the same whole case here: 96x24x12 in all
    assert "96x24x12" in d["config"]["workload"]
    # what the exchanges cost, slowest rank: a halo exchange per sweep, an all-reduce per step
    ex = d["config"]["exchange"]
    assert ex["halo_exchanges"] == 8 * 2 and ex["allreduces"] >= 2
    assert ex["halo_exchange_ms_per_sweep"] > 0 and ex["allreduce_ms_per_step"] > 0
    if scaling == "weak":
        # the strong-scaling leg beside the weak headline: one fixed case on the same ranks
        st = d["config"]["strong"]
        assert "error" not in st, st
        assert st["value"] > 0 and st["particles"] == d["config"]["particles"] and st["exchange"]["halo_exchanges"] == 16
        assert st["single_gpu_value"] is None and st["efficiency"] is None  # no committed single-GPU line of this size
    else:
        assert "strong" not in d["config"]


@pytest.mark.parametrize("how", ["no such case", "out of time"])
def test_a_failing_strong_leg_is_reported_inside_the_line(how):
    # the strong-scaling leg runs in processes of its own: when it cannot be set up, or does not come
    # back in time and is stopped, the headline is still printed, with what went wrong, and the run
    # ends well
    env = dict(os.environ, SPH_TRANSPORT="host")
    extra = ["--fluid", "48x24x12", "--strong-steps", "2"]
    if how == "out of time":
        env["SPH_STRONG_LIMIT_S"] = "0.2"
        extra += ["--strong-fluid", "96x24x12"]
    else:
        extra += ["--strong-fluid", "none"]  # every rank of the leg fails on it
    r = _two_ranks(extra, env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = _check(lines[0], 2, 2, 1, "weak")
    st = d["config"]["strong"]
    assert "error" in st and ("did not finish" in st["error"]) == (how == "out of time"), st


def test_more_ranks_than_gpus_are_refused_before_anything_runs():
    # RCCL refuses two ranks on one GPU: unless the host-staged transport is asked for, the run
    # says so and ends before a GPU is touched -- no line, a failure code
    env = {k: v for k, v in os.environ.items() if k != "SPH_TRANSPORT"}
    r = _two_ranks(["--fluid", "48x24x12"], env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0 and not lines
    assert "one rank per GPU" in r.stderr
