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
    # below 32 images per GPU the trunk passes replay from graphs and every 4th is launched directly and timed: the line
    # still carries the trunk's roofline
    assert d["roofline"] is not None and d["roofline"]["launches"] > 0


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


# ---- bench.py's multi-rank weighting, numerically ------------------------------------------------
def _dp_worker(rank, world, port, outdir, B, V):
    """One rank of a two-rank job on device 0 over gloo: bench.py's own rank_shard() + DataParallelAdam (HIP pack ->
    all-reduce -> unpack -> fused clamp+Adam), decoder only (seeded features stand in for the trunk)."""
    import random
    import torch
    import torch.distributed as dist
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    import capnet  # noqa: F401
    from capnet import ops, synthetic
    from capnet.model import DecoderFactoredLSTM
    from capnet.parallel import DataParallelAdam
    from capnet.train import _backward
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    _, captions, lengths, loss_scale, global_steps = bench.rank_shard(B, V, rank, world, images=False)
    dec = DecoderFactoredLSTM(300, 512, 512, V, 1, dropout=0.0)
    dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=1234))
    dec.to(dev).train()
    feats = torch.randn(B, 300, generator=torch.Generator().manual_seed(500 + rank)).to(dev)
    opt = DataParallelAdam(dec.parameters(), lr=2e-4, overlap=False)
    assert opt.reducer is not None
    random.seed(0)
    tf = [random.random() < 0.8 for _ in range(global_steps)]
    cap_d = captions.to(dev)
    out = dec(cap_d, lengths, feats, tf_mask=tf)
    loss = ops.cross_entropy(out, ops.packed_targets(cap_d, lengths))
    dec.zero_grad()
    _backward(loss, loss_scale)
    opt.step()                     # no clamp pending: .grad afterwards is the reduced gradient
    torch.cuda.synchronize()
    ops.check_device_errors()
    torch.save({"grads": {k: p.grad.cpu() for k, p in dec.named_parameters() if p.grad is not None},
                "params": {k: p.detach().cpu() for k, p in dec.named_parameters()},
                "loss": float(loss.item()), "loss_scale": loss_scale, "tf": tf, "lengths": lengths},
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_is_the_token_weighted_mean(tmp_path):
    """the reduced gradient equals the N-weighted mean of the per-shard reference gradients, and every
    rank holds the same parameters after the update. Two ranks on this box's one GPU over gloo (RCCL refuses two ranks
    per device) through bench.py's own shard / loss_scale / global_steps code; the expectation is built from the
    shards' ACTUAL batches (make_batch with images), so a rank_shard that rebuilds other ranks' lengths from different
    draws (the bug behind commit 3464b6a) fails here."""
    import socket
    import torch
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from capnet import synthetic
    from oracle import decoders_ref as D
    from oracle import step_ref as S
    B, V, world = 64, 8192, 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path), B, V), nprocs=world, join=True)
    got = [torch.load(str(tmp_path / ("rank%d.pt" % r)), weights_only=True) for r in range(world)]
    torch.set_num_threads(16)
    from capnet.model import DecoderFactoredLSTM
    p = synthetic.decoder_state(DecoderFactoredLSTM(300, 512, 512, V, 1).state_dict(), seed=1234)
    shards_full = [synthetic.make_batch(B, V, seed=r)[1:] for r in range(world)]      # the TRUE batches (images drawn)
    n_global = sum(sum(sh[1]) for sh in shards_full)
    want = None
    for r in range(world):
        captions, lengths = shards_full[r]
        assert lengths == got[r]["lengths"]
        assert abs(got[r]["loss_scale"] - sum(lengths) / n_global) < 1e-12
        feats = torch.randn(B, 300, generator=torch.Generator().manual_seed(500 + r))
        tf = got[r]["tf"]
        assert tf == got[0]["tf"]
        loss, grads, _, _ = S.decoder_loss_and_grads(D.factored_lstm_forward, p, captions, lengths, feats,
                                                     tf[:max(lengths)], mode="factual")
        assert abs(got[r]["loss"] - float(loss)) < 1e-4 * float(loss)
        w = sum(lengths) / n_global
        if want is None:
            want = {k: g * w for k, g in grads.items() if g is not None}
        else:
            for k, g in grads.items():
                if g is not None:
                    want[k] += g * w
    for k, g in want.items():
        for r in range(world):
            err = (got[r]["grads"][k] - g).abs().max().item()
            assert err <= 2e-5 * max(g.abs().max().item(), 1e-8) + 1e-9, (k, r, err)
    for k in got[0]["params"]:
        assert torch.equal(got[0]["params"][k], got[1]["params"][k]), k
