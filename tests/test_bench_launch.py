"""bench.py's launcher: `python bench.py --gpus N` must start its N ranks itself when no launcher set WORLD_SIZE (the
driver runs it exactly like the N = 1 line), run the chunk-parallel exchange, and print ONE JSON line from rank 0.

CPU: the self-launch path is taken and fails loudly per rank (no GPU -> no product path).  GPU: two ranks on ONE GPU
through the `VIDTOME_BENCH_BACKEND=gloo` hook in all three exchange modes; with two or more GPUs also over RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the CPU half of the launcher test")
def test_self_launch_without_gpu_fails_loudly_per_rank():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=300)
    assert p.returncode != 0
    assert "AssertionError" not in p.stderr
    assert p.stderr.count("needs an MI355X") >= 1          # a rank that died takes the other down by PID
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error_not_an_assert():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=300)
    assert p.returncode != 0 and "AssertionError" not in p.stderr
    assert ("WORLD_SIZE=1" in p.stderr) or ("needs an MI355X" in p.stderr)


def _line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["neighbour", "allgather", "ring"])
def test_bench_two_ranks_one_gpu_gloo(exchange):
    """The N > 1 branch end to end (self-launch -> process group -> exchange through compute_merge -> barrier-fenced
    timing -> per-rank times -> one JSON line), two ranks sharing cuda:0 over gloo."""
    d = _line(_run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--exchange", exchange,
                    "--exchange-modes", "none"], {"VIDTOME_BENCH_BACKEND": "gloo"}))
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["launcher"] == "self" and d["backend"] == "gloo"
    assert len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert abs(d["ms_per_step"] - max(d["per_rank_ms_per_step"])) < 1e-3
    assert d["roofline"]["achieved"] > 0 and d["matching"]["calls"] > 0
    assert d["config"]["exchange"] == exchange and "chunk-parallel x2" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d and "exchange_modes" not in d and "regimes" not in d


@pytest.mark.gpu
def test_bench_two_ranks_report_every_exchange_mode():
    """The default N > 1 line: the headline's mode (neighbour) plus the other two modes timed in the same process group --
    what the first 8-GPU run needs for the ring-vs-neighbour comparison."""
    d = _line(_run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {"VIDTOME_BENCH_BACKEND": "gloo"},
                   timeout=1200))
    assert d["config"]["exchange"] == "neighbour"
    modes = d["exchange_modes"]
    assert set(modes) == {"neighbour", "ring", "allgather"}
    for m, r in modes.items():
        assert "error" not in r, (m, r)
        assert r["ms_per_step"] > 0 and len(r["per_rank_ms_per_step"]) == 2
        assert r["exchange_bytes_per_step"]["sent"] > 0
    assert abs(modes["neighbour"]["ms_per_step"] - d["ms_per_step"]) < 1e-3


@pytest.mark.gpu
def test_bench_four_ranks_one_gpu_gloo():
    """World size 4 (one process group per directed ring edge, no two-rank special case), neighbour exchange with the
    early hand-over, the four ranks sharing cuda:0 over gloo."""
    d = _line(_run(["--gpus", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--exchange-modes", "none"],
                   {"VIDTOME_BENCH_BACKEND": "gloo"}, timeout=1200))
    assert d["n_gpus"] == 4 and d["ranks"] == 4 and len(d["per_rank_ms_per_step"]) == 4
    assert d["config"]["exchange"] == "neighbour"
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
@pytest.mark.parametrize("exchange", ["neighbour", "allgather", "ring"])
def test_bench_two_ranks_rccl(exchange):
    d = _line(_run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--exchange", exchange]))
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"].startswith("rccl")
    assert len(d["per_rank_ms_per_step"]) == 2


@pytest.mark.gpu
def test_bench_single_gpu_line_has_the_contract_fields():
    d = _line(_run(["--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--regimes", "n01,corr05", "--regime-steps", "2"]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["ranks"] == 1 and d["roofline"]["event_passes"] == 1
    assert "different chunk" in d["config"]["regime"]
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # SURVEY 8d's regimes ride on the line; the headline is the harder of the two it names
    assert d["config"]["data_regime"].startswith("corr01") and d["config"]["headline_configuration"] is True
    assert set(d["regimes"]) == {"corr01", "n01", "corr05"}
    assert abs(d["regimes"]["corr01"]["ms_per_step"] - d["ms_per_step"]) < 1e-3
    for name, e in d["regimes"].items():
        assert e["ms_per_step"] > 0 and e["matching_ms"] > 0 and e["attention_ms"] > 0 and e["pairs_per_row"] > 0, (name, e)
        assert 0.0 <= e["pruned_block_fraction"] <= 1.0 and 0.0 < e["executed_mfma_fraction"] <= 1.0, (name, e)
    assert d["regimes"]["corr05"]["vs_corr05"] == 1.0
    # the matcher's flops are nominal (the filter prunes); the pruned fraction is a device-side counter
    m = d["matching"]
    assert "nominal_tflops" in m and "executed_tflops" not in m and m["counters"]["blocks_tested"] > 0
    assert d["regimes"]["n01"]["pruned_block_fraction"] < d["regimes"]["corr05"]["pruned_block_fraction"]


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["cfg3", "cfg5"])
def test_bench_secondary_workloads(workload):
    d = _line(_run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--regimes", "none", "--workload", workload],
                   timeout=1200))
    assert d["config"]["headline_configuration"] is False and "SECONDARY" in d["metric"]
    assert d["config"]["batch"] == (3 if workload == "cfg3" else 2)
    assert d["roofline"]["achieved"] > 0 and d["matching"]["matching_ms_per_step"] > 0
    assert d["components_ms_per_step"]["attention"] > 0
