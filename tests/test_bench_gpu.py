"""bench.py end to end: the one-line JSON contract on one rank, and a two-rank rehearsal of the
multi-GPU launch (`torch.distributed.run`, rank-local shards, loss scaling, barrier / max-over-ranks
timing) with both ranks on this box's single GPU over gloo -- RCCL refuses two ranks on one device,
so the collective itself is covered by CAPNET_FORCE_ALLREDUCE (tests/test_step_gpu.py) and the CPU
gloo tests (tests/test_parallel_cpu.py). Children are separate processes; nothing is exec-replaced."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra, timeout=600):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d: %r" % (len(lines), p.stdout[:400])
    return json.loads(lines[0])


def test_bench_prints_one_json_line():
    d = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], {})
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "images/sec" and d["scaling"] == "weak"
    assert d["value"] > 0 and math.isfinite(d["loss_last"]) and d["dtype"].startswith("f32 (")
    r = d["roofline"]
    # `peak` is the rate at which both matrix pipes, each at its own peak, get through what the kernels issue
    # on them (Winograd 16/36 of the direct sum on the f32 pipe, the split 1x1 convs 3 f16 / 6 bf16 products per
    # multiply on the 16-bit pipe), so `frac` = time at peak / conv time is bounded by 1 and equals executed.frac
    e = r["executed"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(e["frac"] - r["frac"]) < 2e-3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] > 157.3
    for pipe in ("f32_mfma", "f16_mfma", "bf16_mfma"):
        assert 0 <= e[pipe]["achieved"] < e[pipe]["peak"]
    assert abs(sum(e[pipe]["share_of_algorithmic_flops"] for pipe in ("f32_mfma", "f16_mfma", "bf16_mfma")) - 1) < 1e-3
    assert r["traffic"] is None or "offline" in r["traffic_source"]
    assert d["roofline_lstm_step"]["bound"] == "hbm"


@pytest.mark.parametrize("extra", [["--decoder", "att", "--batch", "12"], ["--decoder", "nic"],
                                   ["--layers", "3", "--factored", "1024", "--batch", "16"]])
def test_secondary_workloads_run(extra):
    """The secondary bench lines of profiles/ (attention decoder at 12 per GPU: the one-launch step products; NIC;
    configs[4]'s stacked decoder shape) start, train and print the contract's line."""
    d = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-lstm-roofline"] + extra, {})
    assert d["value"] > 0 and math.isfinite(d["loss_first"]) and math.isfinite(d["loss_last"])


def test_two_rank_rehearsal_on_one_gpu():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-lstm-roofline"]
    d = _run(cmd, {"CAPNET_REHEARSE_ONE_GPU": "1", "MASTER_ADDR": "127.0.0.1"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and math.isfinite(d["loss_first"]) and math.isfinite(d["loss_last"])
    assert d["cpu_baseline"] is None


def test_self_launched_two_rank_rehearsal():
    """The driver's command shape without a launcher: `python3 bench.py --gpus 2` starts its own two
    rank processes before touching the GPU and relays rank 0's line (rehearsed on device 0 / gloo)."""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--no-lstm-roofline"], {"CAPNET_REHEARSE_ONE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and math.isfinite(d["loss_last"]) and d["roofline"]["frac"] > 0


def test_self_launch_propagates_a_rank_failure():
    env = dict(os.environ)
    env["CAPNET_REHEARSE_ONE_GPU"] = "1"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--decoder", "factored", "--batch", "0"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
