#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full StyleNet train step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = ResNet-152 trunk forward (train-mode BN, no_grad) + encoder head + FactoredLSTM-512
decoder forward (teacher forcing 0.8, dropout 0.5) + softmax-NLL + backward + element-wise clamp
+ Adam, on a batch of 64 synthetic 224x224 images per GPU (BASELINE.json configs[1]; weak
scaling for N > 1 with one RCCL all-reduce of the flat gradient buffer per step). Inputs are
resident in HBM before the timed region. Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix peak (dense)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: BF16 MFMA, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--vocab", type=int, default=8192)
    ap.add_argument("--layers", type=int, default=1,
                    help="stacked FactoredLSTM layers (configs[3] / [4]: 2 / 3). PERF-ONLY, PARITY UNPINNED: the reference "
                         "ignores num_layers; capnet.stacked defines the stacking (SURVEY App. A-1)")
    ap.add_argument("--factored", type=int, default=512, help="factored size (configs[4]: 1024)")
    ap.add_argument("--decoder", default="factored", choices=["factored", "nic", "att"],
                    help="factored = BASELINE configs[1] (the headline); nic = config 0's decoder; "
                         "att = config 3's attention decoder (secondary workloads)")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-events", action="store_true",
                    help="do not bracket conv kernels with hipEvents (roofline becomes null)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run trunk and decoder of a step back to back on one stream instead of "
                         "overlapping step i's decoder with step i+1's trunk (capnet.train.TrunkPipeline)")
    ap.add_argument("--conv-event-every", type=int, default=0,
                    help="bracket the conv launches of every N-th trunk pass only (N > 1: cheaper, but the passes in flight "
                         "beside a timed one are then not in the union of intervals: per-launch times under contention). "
                         "Default: 1 for batches of at least 32 images per GPU (the headline), 4 below -- such a step is "
                         "bound by launch chains and the host, where 620 event calls per pass make the timing itself the "
                         "largest and least stable cost (12 images, attention decoder: 4.3-5.4 ms per step with N = 1, "
                         "4.09 +- 0.01 with N = 4)")
    ap.add_argument("--graph-trunk", action="store_true",
                    help="replay the trunk passes from hipGraphs (the passes whose conv launches are bracketed by events are "
                         "launched directly). Default below 32 images per GPU, where the step is bound by the host's launch "
                         "calls (2 images: 3.9 ms per step, 2.9 with the trunk replayed)")
    ap.add_argument("--no-graph-trunk", action="store_true")
    ap.add_argument("--pipeline-depth", type=int, default=3,
                    help="trunk passes in flight ahead of the decoder")
    ap.add_argument("--no-lstm-roofline", action="store_true",
                    help="skip the LSTM-step microbenchmark (PMC passes profile the train step only)")
    ap.add_argument("--cpu-steps", type=int, default=5)
    args = ap.parse_args()
    if args.conv_event_every <= 0:
        args.conv_event_every = 1 if args.batch >= 32 else 4
    if args.batch < 32 and not args.no_graph_trunk:
        args.graph_trunk = True
    return args


def pmc_traffic(key="conv_bytes_per_launch"):
    """(bytes per launch, provenance) of HBM traffic from the TCC counters (FETCH_SIZE x2 on gfx950
    + WRITE_SIZE). PMC passes cannot run inside the timed bench (one counter per pass, rocprofv3
    attached): the value is measured OFFLINE by tools/pmc_traffic.sh on this same command and read
    from the newest profiles/round*_pmc_traffic.json; (None, None) if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
            if key in d:
                return d[key], ("offline: %s (tools/pmc_traffic.sh -> tools/pmc_summarize.py; rocprofv3 "
                                "--kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `%s`), "
                                "not measured by this run" % (os.path.relpath(path, ROOT),
                                                            d.get("command", "bench.py").split("-- ")[-1]))
        except Exception:
            continue
    return None, None


def log(msg):
    sys.stderr.write("[bench %s] %s\n" % (time.strftime("%H:%M:%S"), msg))
    sys.stderr.flush()


def host_cores():
    """CPU share of this process: min(affinity, cgroup quota); the GPU boxes give 16 per GPU."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(args, steps, warmup=2, parity=None):
    """The CPU oracle (oracle/: torch-CPU restatement of the reference step) timed on this host's
    cores, same workload as the GPU line: batch 64, ResNet-152 train-mode trunk + head +
    FactoredLSTM-512, V=8192, tf 0.8, dropout as on the GPU (a host-drawn mask). SURVEY 8(d) /
    BASELINE.md 3 procedure: `warmup` untimed + `steps` (>= 5) timed steps, median; the trunk and
    the trainable half (head + decoder forward, loss, backward, clamp, Adam) timed separately."""
    from capnet import synthetic
    from oracle import decoders_ref as D
    from oracle import step_ref as S
    from oracle.resnet152_ref import EncoderCNNRef
    import torch.nn.functional as Fn
    cores = host_cores()
    torch.set_num_threads(cores)
    log("cpu baseline on %d threads" % cores)
    B, V = args.batch, args.vocab
    enc = EncoderCNNRef(300)
    enc.train()
    from capnet.model import DecoderFactoredLSTM
    shapes = {k: v for k, v in DecoderFactoredLSTM(300, 512, 512, V, 1).state_dict().items()}
    p = synthetic.decoder_state(shapes, seed=1234)
    imgs, captions, lengths = synthetic.make_batch(B, V, seed=0)
    opt = S.AdamRef(lr=2e-4)
    random.seed(0)
    g = torch.Generator().manual_seed(0)
    keep = 1.0 - args.dropout
    t_all, t_trunk, losses = [], [], []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        tf = [random.random() < 0.8 for _ in range(max(lengths))]
        drop = None
        if args.dropout > 0:      # nn.Dropout on the embedded captions (stylenet/model.py:166-167)
            drop = (torch.rand(captions.shape + (300,), generator=g) < keep).float() / keep
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
        with torch.no_grad():
            pooled = enc.resnet(imgs).reshape(B, -1)          # frozen trunk (model.py:23-25)
        t1 = time.perf_counter()
        feats = enc.bn(enc.linear(pooled))
        logits = D.factored_lstm_forward(leaves, captions, lengths, feats, tf, "factual", drop_mask=drop)
        loss = Fn.cross_entropy(logits, D.packed_targets(captions, lengths))
        enc.zero_grad()
        loss.backward()
        grads = {k: v.grad for k, v in leaves.items()}
        S.clip_gradient_(grads.values(), 0.5)
        with torch.no_grad():
            opt.step(p, grads)
            hp = {("enc." + k): v for k, v in enc.named_parameters() if not k.startswith("resnet.")}
            hg = {k: v.grad for k, v in hp.items()}
            S.clip_gradient_([g_ for g_ in hg.values() if g_ is not None], 0.5)
            opt.step(hp, hg)
        t2 = time.perf_counter()
        log("cpu step %d: %.2f s (trunk %.2f s), loss %.5f" % (it, t2 - t0, t1 - t0, float(loss)))
        if it >= warmup:
            t_all.append(t2 - t0)
            t_trunk.append(t1 - t0)
        losses.append(float(loss))
    med = lambda v: sorted(v)[len(v) // 2]
    t, tt = med(t_all), med(t_trunk)
    if parity is not None:
        parity.update(oracle_parity_losses(parity, S, D, EncoderCNNRef, Fn))
    return {"value": round(B / t, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d full train steps at batch %d after %d warm-up steps (median %.2f s/step: trunk "
                      "%.2f s + head/decoder/loss/backward/clamp/Adam %.2f s), oracle/ torch-CPU fp32, "
                      "dropout %.2f" % (steps, B, warmup, t, tt, t - tt, args.dropout),
            "trunk_s": round(tt, 3), "decoder_s": round(t - tt, 3), "step_s": round(t, 3),
            "loss_first": losses[0]}


PARITY_STEPS, PARITY_LR, PARITY_TF_SEED = 2, 2e-3, 3


def parity_inputs(B, V):
    """Inputs of the untimed loss-parity leg: BASELINE configs[1] (batch 64, V 8192), seeded weights on both sides
    (capnet.synthetic), dropout 0 (the GPU's dropout stream is its own), one fixed teacher-forcing mask per step."""
    from capnet import synthetic
    imgs, captions, lengths = synthetic.make_batch(B, V, seed=0)
    random.seed(PARITY_TF_SEED)
    tfs = [[random.random() < 0.8 for _ in range(max(lengths))] for _ in range(PARITY_STEPS)]
    return imgs, captions, lengths, tfs


def gpu_parity_losses(dev, B, V):
    """The PRODUCT side of `loss_parity`: PARITY_STEPS whole train steps (stylenet/train_multitask.py:373-389: trunk,
    head, decoder, NLL, backward, clamp, Adam) through capnet's sequential loop entry on fresh modules."""
    from capnet import ops, synthetic
    from capnet.model import DecoderFactoredLSTM, EncoderCNN
    from capnet.optim import Adam
    from capnet.train import CrossEntropyLoss, train_step
    enc = EncoderCNN(300)
    est = synthetic.encoder_state(enc.state_dict(), seed=1234)
    enc.load_state_dict(est)
    dec = DecoderFactoredLSTM(300, 512, 512, V, 1, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=1234)
    dec.load_state_dict(p)
    enc.to(dev).train()
    dec.to(dev).train()
    imgs, captions, lengths, tfs = parity_inputs(B, V)
    opt = Adam(list(dec.parameters()) + list(enc.linear.parameters()) + list(enc.bn.parameters()), lr=PARITY_LR)
    imgs_d, caps_d = imgs.to(dev), captions.to(dev)
    got = [float(train_step(enc, dec, opt, CrossEntropyLoss(), imgs_d, caps_d, lengths, 0.5, tf_mask=tfs[it]).item())
           for it in range(PARITY_STEPS)]
    ops.check_device_errors()
    return {"B": B, "V": V, "gpu": got, "est": est, "p": p}


def oracle_parity_losses(parity, S, D, EncoderCNNRef, Fn):
    """The CHECKER side of `loss_parity`, run inside the cpu_baseline leg (the only place bench.py touches oracle/):
    the same steps through the CPU restatement, fp32."""
    B, V = parity["B"], parity["V"]
    imgs, captions, lengths, tfs = parity_inputs(B, V)
    ref_enc = EncoderCNNRef(300)
    ref_enc.load_state_dict({k: v.clone() for k, v in parity.pop("est").items()})
    ref_enc.train()
    p_ref = {k: v.clone() for k, v in parity.pop("p").items()}
    opt_ref = S.AdamRef(lr=PARITY_LR)
    out = []
    for it in range(PARITY_STEPS):
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p_ref.items()}
        feats = ref_enc(imgs)
        logits = D.factored_lstm_forward(leaves, captions, lengths, feats, tfs[it], "factual")
        loss = Fn.cross_entropy(logits, D.packed_targets(captions, lengths))
        ref_enc.zero_grad()
        loss.backward()
        grads = {k: v.grad for k, v in leaves.items()}
        S.clip_gradient_(grads.values(), 0.5)
        with torch.no_grad():
            hp = {("enc." + k): v for k, v in ref_enc.named_parameters() if not k.startswith("resnet.")}
            hg = {k: v.grad for k, v in hp.items()}
            S.clip_gradient_([g_ for g_ in hg.values() if g_ is not None], 0.5)
            both, both_g = dict(p_ref), dict(grads)
            both.update(hp)
            both_g.update(hg)
            opt_ref.step(both, both_g)
        out.append(float(loss.detach()))
        log("loss parity step %d: oracle %.6f, gpu %.6f" % (it, out[-1], parity["gpu"][it]))
    rel = [abs(a - b) / abs(b) for a, b in zip(parity["gpu"], out)]
    return {"oracle": out, "rel": [float("%.3e" % r) for r in rel], "rel_max": float("%.3e" % max(rel)),
            "steps": PARITY_STEPS, "tolerance": 1e-4,
            "what": "whole train step (ResNet-152 train-mode trunk + head + FactoredLSTM-512 + NLL + backward + clamp 0.5 "
                    "+ Adam %g), batch %d, V %d, dropout 0, tf 0.8 with a fixed mask per step (random.seed(%d)), seeded "
                    "weights on both sides; gpu = capnet.train.train_step through libcapnet_hip.so, oracle = oracle/ "
                    "torch-CPU fp32; untimed" % (PARITY_LR, B, V, PARITY_TF_SEED)}


def rank_shard(B, V, rank, world, images=True):
    """(images, captions, lengths, loss_scale, global_steps) of `rank`'s shard of the weak-scaling job: B seeded
    (image, caption) pairs per rank. Every rank can rebuild every rank's lengths (seeded), so the token weights and the
    global number of decoder steps need no communication. loss_scale = N_rank / N_global: with a SUM all-reduce of the
    gradients that is the gradient of the global token-mean loss; global_steps = the longest caption of the
    job, so that every rank consumes the same teacher-forcing draws.
    tests/test_bench_gpu.py::test_two_rank_gradient_is_the_token_weighted_mean holds this very function, with
    capnet.parallel.DataParallelAdam behind it, to the per-shard oracle gradients."""
    from capnet import synthetic
    all_lengths = [synthetic.make_batch(B, V, seed=r, images=False)[2] for r in range(world)]
    n_global = sum(sum(l) for l in all_lengths)
    global_steps = max(l[0] for l in all_lengths)
    imgs, captions, lengths = synthetic.make_batch(B, V, seed=rank, images=images)
    if lengths != all_lengths[rank]:
        raise RuntimeError("synthetic.make_batch: lengths of a seed depend on the images flag")
    loss_scale = float(sum(lengths)) / n_global if world > 1 else None
    return imgs, captions, lengths, loss_scale, global_steps


def self_launch(args):
    """`python3 bench.py --gpus N` without a launcher: start N rank processes of this script (one
    per GPU, env-style rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU,
    relay rank 0's JSON line, fail if any rank fails. The parent never initialises HIP and never
    exec-replaces itself."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus),
                    "LOCAL_WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]
    if any(rcs) or not lines:
        sys.stderr.write("[bench] rank exit codes %s, %d JSON line(s)\n" % (rcs, len(lines)))
        raise SystemExit(next((rc for rc in rcs if rc), 1))
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to fd 1 when the
    # process group comes up, so everything else this process (and its libraries) writes to stdout
    # goes to stderr, and the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    # CAPNET_REHEARSE_ONE_GPU=1: every rank on device 0 with the gloo backend -- a multi-rank
    # rehearsal of this script's sharding / scaling / barrier logic on a one-GPU box (RCCL refuses
    # two ranks on one device). Not a measurement.
    rehearse = os.environ.get("CAPNET_REHEARSE_ONE_GPU") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        # the ranks share the host: keep torch's CPU pool small (the step runs no CPU operator)
        torch.set_num_threads(max(1, min(4, host_cores())))
    import torch.distributed as dist
    if world > 1 or os.environ.get("CAPNET_FORCE_ALLREDUCE") == "1":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import capnet
    from capnet import ops, synthetic
    from capnet.model import DecoderFactoredLSTM, EncoderCNN
    from capnet.nic_model import DecoderRNN
    from capnet.parallel import DataParallelAdam
    from capnet.train import CrossEntropyLoss, TrunkPipeline, train_step, train_step_att
    from capnet import model_att

    B, V = args.batch, args.vocab
    torch.manual_seed(1234)
    encoder = EncoderCNN(300) if args.decoder != "att" else model_att.EncoderCNN(14)
    if args.decoder == "factored" and args.layers > 1:
        from capnet.stacked import StackedFactoredLSTM
        decoder = StackedFactoredLSTM(300, 512, args.factored, V, args.layers, dropout=args.dropout)
    elif args.decoder == "factored":
        decoder = DecoderFactoredLSTM(300, 512, args.factored, V, 1, dropout=args.dropout)
    elif args.decoder == "att":
        decoder = model_att.DecoderFactoredLSTMAtt(512, 300, 512, 512, V, 1, dropout=args.dropout)
    else:
        decoder = DecoderRNN(300, 512, V, 1, dropout=args.dropout)
    decoder.load_state_dict(synthetic.decoder_state(decoder.state_dict(), seed=1234))
    encoder.to(dev).train()
    decoder.to(dev).train()
    params = list(decoder.parameters())
    if args.decoder != "att":
        params += list(encoder.linear.parameters()) + list(encoder.bn.parameters())
    # pipelined: everything trainable already runs on the pipeline's side stream, in order
    optimizer = DataParallelAdam(params, lr=2e-4, overlap=args.no_pipeline)
    if args.decoder != "att":
        optimizer.attach(encoder)
    criterion = CrossEntropyLoss()

    images, captions, lengths, loss_scale, global_steps = rank_shard(B, V, rank, world)
    images, captions = images.to(dev), captions.to(dev)
    random.seed(0)

    pipe = None
    if not args.no_pipeline:
        pipe = TrunkPipeline(encoder, decoder, optimizer, criterion, 0.5, attention=args.decoder == "att",
                             depth=args.pipeline_depth, graph_trunk=args.graph_trunk)
        for _ in range(pipe.depth):
            pipe.prefetch(images)

    def step():
        tf = [random.random() < 0.8 for _ in range(global_steps)]
        if pipe is not None:
            # the trunk of the next batch goes to the main stream first, then this batch's
            # head + decoder + loss + backward + clamp + Adam to the side stream
            return pipe.step(captions, lengths, next_images=images, tf_mask=tf, loss_scale=loss_scale)
        if args.decoder == "att":
            optimizer.wait_for_update()   # no trainable encoder head to hang the wait on
            return train_step_att(encoder, decoder, optimizer, criterion, images, captions, lengths,
                                  0.5, tf_mask=tf, loss_scale=loss_scale)
        return train_step(encoder, decoder, optimizer, criterion, images, captions, lengths, 0.5,
                          tf_mask=tf, loss_scale=loss_scale)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    log("rank %d/%d ready: batch %d, %d tokens, %d steps" % (rank, world, B, sum(lengths), lengths[0]))
    first = None
    for _ in range(args.warmup):
        l = step()
        first = l if first is None else first
    runner = encoder._trunk()
    plan = runner._plan(B, 224, 224, dev)
    lib = capnet.lib()
    # two hipEvents per conv launch: rank 0 only (the roofline is rank 0's; the other ranks share
    # the host with it and need not pay for 6 200 event calls per 20 steps)
    conv_events = not args.no_conv_events and rank == 0
    if conv_events:
        # 310 event records per pass are bubbles in the pass's stream: they cost 2.5 % images/s (7 480 vs 7 680 without)
        runner.set_timing(plan, args.conv_event_every)
    barrier()
    log("warm-up done, timing %d steps" % args.steps)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    host_enqueue = time.perf_counter() - t0     # host time to queue the steps (GPU may lag behind)
    optimizer.wait_for_update()
    barrier()
    elapsed = time.perf_counter() - t0
    log("host enqueue %.3f s of %.3f s" % (host_enqueue, elapsed))
    log("timed region: %.3f s" % elapsed)
    runner.set_timing(plan, False)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ops.check_device_errors()

    roofline = None
    if conv_events:
        ms, n, fl = C.c_double(), C.c_long(), C.c_double()
        capnet._lib.check(lib.capnet_trunk_collect_timing(plan["handle"], C.byref(ms), C.byref(n), C.byref(fl)))
        if n.value > 0:
            achieved = fl.value / (ms.value * 1e-3) / 1e12
            # What the matrix pipes are asked to issue for one pass. Weight image kind 2 = Winograd: 16 multiplies
            # per 2x2 outputs and input channel instead of 36, f32 MFMA. Kind 5 = split-f16 1x1 conv: three f16
            # products per multiply; kind 4 = split-bf16: six bf16 products (both on the 16-bit matrix pipe, peak
            # 2.5 PFLOP/s dense). Everything else: f32 MFMA as counted. The peak the union of the conv launches is
            # priced against is the rate at which BOTH pipes, each at its own peak, would get through that work:
            # algorithmic flops / (f32-issued / 157.3 + 16-bit-issued / 2500).
            algo = f32x = bf16x = f16x = algo_f16 = 0.0
            for i in range(lib.capnet_trunk_num_convs(plan["handle"])):
                fi = lib.capnet_trunk_conv_flops(plan["handle"], i)
                kind = lib.capnet_trunk_conv_kmajor(plan["handle"], i)
                algo += fi
                if kind in (5, 7, 8):
                    f16x += 3.0 * fi
                    algo_f16 += fi
                elif kind == 6:      # the stem: K = 147 issued as 22 rows x 8 taps = 176
                    f16x += 3.0 * fi * 176.0 / 147.0
                    algo_f16 += fi
                else:
                    f32x += fi
            t_peak = f32x / MFMA_F32_PEAK_TFLOPS + (bf16x + f16x) / MFMA_BF16_PEAK_TFLOPS     # per pass, in 1e-12 s
            peak_equiv = algo / t_peak
            per_s = achieved / algo          # TFLOP/s per algorithmic flop of a pass
            roofline = {"bound": "mfma",
                        "kernel": "the trunk's 155 convolutions, all on three f16 MFMA products of 2-way split fp32 operands per "
                                  "multiply (fp32-grade results): fb_fused_kernel (the 44 block boundaries inside stages 1-3: conv3 + "
                                  "bn3 + residual + ReLU + the next conv1 in one launch that never forms y3; its BatchNorm statistics "
                                  "come from the Gram matrix of conv3's input -- fb_gram / fb_gram_reduce / fb_quad launches, timed here "
                                  "with zero flops; CAPNET_NO_FUSED_BLOCK=1 restores the y3 data flow), conv3x3_patch_kernel (the 47 "
                                  "stride-1 3x3 convs), conv_f16x3_kernel (strided 3x3, downsample, first conv1, conv3 at stage "
                                  "transitions / stage 4), conv1x1_tail_kernel (the 5 conv1 that absorb a materialised tail), "
                                  "conv_stem_f16x3_kernel (K = 147 issued as 176); CAPNET_NO_H3=1 puts everything on the f32-MFMA kernels",
                        "flops": "ALGORITHMIC: 2*M*Cout*KH*KW*Cin of the direct sum (SURVEY 8d: 23.02 GFLOP per image)",
                        "peak_is": "algorithmic flops / (time the f32 matrix pipe needs for what is issued on it at "
                                   "157.3 TFLOP/s + time the 16-bit matrix pipe needs for its share at 2500 TFLOP/s): "
                                   "a split-f16 conv issues 3 f16 products per multiply (the stem x 176/147 for its padded K), "
                                   "anything planned for the f32-MFMA kernels as counted (breakdown in `executed`); "
                                   "`frac_algorithmic_vs_f16_dense` prices the ALGORITHMIC flops against the 2 500 TFLOP/s "
                                   "dense f16 peak instead (SURVEY 8d's reading: the three products count as overhead)",
                        "how": "HIP events around every conv launch of every %d-th trunk pass of the timed region, on its "
                               "launch stream; " % args.conv_event_every +
                               "duration = time with at least one conv launch running (union of the "
                               "intervals: three trunk passes are in flight, their launches overlap)",
                        "achieved": round(achieved, 2), "peak": round(peak_equiv, 2), "unit": "TFLOP/s",
                        "frac": round(achieved / peak_equiv, 4),
                        "frac_algorithmic_vs_f16_dense": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                        "traffic": pmc_traffic()[0],
                        "traffic_source": pmc_traffic()[1],
                        "launches": n.value, "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
                        "flops_per_launch": fl.value / n.value,
                        "vs_f32_matrix_peak": round(achieved / MFMA_F32_PEAK_TFLOPS, 4)}
            roofline["executed"] = {
                "f32_mfma": {"achieved": round(f32x * per_s, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "share_of_algorithmic_flops": round((algo - bf16x / 6.0 - algo_f16) / algo, 4)},
                "f16_mfma": {"achieved": round(f16x * per_s, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "share_of_algorithmic_flops": round(algo_f16 / algo, 4)},
                "bf16_mfma": {"achieved": round(bf16x * per_s, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "share_of_algorithmic_flops": round(bf16x / 6.0 / algo, 4)},
                "frac": round(f32x * per_s / MFMA_F32_PEAK_TFLOPS + (bf16x + f16x) * per_s / MFMA_BF16_PEAK_TFLOPS, 4),
                "note": "flops the matrix pipes issued per second of conv time; frac = sum over the pipes of issued / "
                        "peak = roofline.frac"}
            if pipe is not None:
                # In the timed region the convolutions share the chip with the previous batch's
                # decoder (that is where the throughput comes from, and it lengthens each conv a
                # little). For reference: the same kernels with nothing else running, 3 extra
                # untimed trunk passes.
                torch.cuda.synchronize()
                runner.set_timing(plan, True)
                for _ in range(3):
                    encoder(images) if args.decoder == "att" else encoder.trunk_features(images)
                torch.cuda.synchronize()
                runner.set_timing(plan, False)
                capnet._lib.check(lib.capnet_trunk_collect_timing(plan["handle"], C.byref(ms), C.byref(n),
                                                                 C.byref(fl)))
                if n.value > 0:
                    alone = fl.value / (ms.value * 1e-3) / 1e12
                    roofline["alone"] = {"achieved": round(alone, 2),
                                         "frac": round(alone / peak_equiv, 4),
                                         "note": "same conv launches without the overlapped decoder"}

    # secondary roofline: the recurrent LSTM step (5.77 MB of algorithmic HBM
    # traffic per step at b=64, H=512), timed with events on the launch stream
    lstm = None
    if rank == 0 and not args.no_lstm_roofline:
        Hh, bb = 512, 64
        hp = torch.randn(bb, Hh, device=dev) * 0.1
        wc = torch.randn(4 * Hh, Hh, device=dev) * 0.05
        wfrag = torch.empty(lib.capnet_lstm_wfrag_floats(Hh), device=dev)
        capnet._lib.check(lib.capnet_lstm_pack_wfrag(wc.data_ptr(), wfrag.data_ptr(), Hh, 0,
                                                     capnet._lib.current_stream()))
        # 24 dependent steps (one caption's worth): step i reads h, c of step i-1 and its own rows
        # of the pre-activation buffer, as in capnet_seq_forward
        n_steps, n_rep = 24, 20
        gts = torch.randn(n_steps, bb, 4 * Hh, device=dev)
        hbuf = [hp, torch.empty_like(hp)]
        cbuf = [torch.randn(bb, Hh, device=dev) * 0.1, torch.empty(bb, Hh, device=dev)]
        st = capnet._lib.current_stream()
        step_no = [0]

        def lstm_step():
            i = step_no[0] % n_steps
            step_no[0] += 1
            capnet._lib.check(lib.capnet_lstm_step_fused(
                hbuf[i % 2].data_ptr(), wfrag.data_ptr(), gts[i].data_ptr(), 4 * Hh,
                cbuf[i % 2].data_ptr(), cbuf[(i + 1) % 2].data_ptr(), hbuf[(i + 1) % 2].data_ptr(),
                bb, Hh, 0, st))
        for _ in range(n_steps):
            lstm_step()
        torch.cuda.synchronize()
        # captured in a hipGraph, so that the host's ctypes/launch cost (~8 us per call) is not
        # what gets timed
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            st = capnet._lib.current_stream()
            with torch.cuda.graph(graph, stream=side):
                st = capnet._lib.current_stream()
                for _ in range(n_steps):
                    lstm_step()
        st = capnet._lib.current_stream()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_rep):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (n_rep * n_steps)
        step_bytes = 4 * Hh * Hh * 4 + bb * 4 * Hh * 4 + bb * Hh * 4 * 4 + bb * 4 * Hh * 4
        per_launch = {"kernel": "lstm_step_fused_kernel, one launch per step replayed from a hipGraph",
                      "us_per_step": round(us, 2), "achieved": round(step_bytes / us / 1e3, 1),
                      "frac": round(step_bytes / us / 1e3 / 8000.0, 4)}
        lstm = {"bound": "hbm", "kernel": "lstm_step_fused_kernel (b=64, H=512; 24 dependent steps replayed from a hipGraph)",
                "achieved": round(step_bytes / us / 1e3, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(step_bytes / us / 1e3 / 8000.0, 4),
                "traffic": pmc_traffic("lstm_step_bytes_per_launch")[0],
                "traffic_source": pmc_traffic("lstm_step_bytes_per_launch")[1],
                "bytes_per_step": step_bytes, "us_per_step": round(us, 2)}
        if lib.capnet_lstm_persist_supported(bb, Hh):
            # the product path for teacher-forced runs: ONE launch for the 24 dependent steps, weights
            # register-resident (csrc/lstm_persist.hip); events on the launch stream around n_rep launches
            img = torch.empty(lib.capnet_lstm_persist_w_floats(), device=dev)
            capnet._lib.check(lib.capnet_lstm_persist_pack(wc.data_ptr(), img.data_ptr(), 0, capnet._lib.current_stream()))
            bs24 = capnet._lib.int_array([bb] * n_steps)
            G24 = torch.randn(n_steps * bb, 4 * Hh, device=dev)
            G0 = G24.clone()
            n_rep = 20
            C24 = torch.empty(n_steps * bb, Hh, device=dev)
            H24 = torch.empty(n_steps * bb, Hh, device=dev)
            # one zeroed control block per launch, so that launches can follow each other without a
            # memset in between and no flag of an earlier launch can satisfy a wait
            ctls = torch.zeros(n_rep, lib.capnet_lstm_persist_ctl_ints(), dtype=torch.int32, device=dev)
            errf = ops.err_flag(dev)

            def persist_burst():
                for k in range(n_rep):
                    capnet._lib.check(lib.capnet_lstm_persist_run(
                        img.data_ptr(), G24.data_ptr(), C24.data_ptr(), H24.data_ptr(), bs24, 0, n_steps, Hh, 0, 1,
                        ctls[k].data_ptr(), errf.data_ptr(), None, capnet._lib.current_stream()))
            bursts = []
            for rep in range(4):
                ctls.zero_()
                G24.copy_(G0)        # (the launch overwrites its pre-activations with the gates: the
                torch.cuda.synchronize()   #  later launches of a burst run on gate values; same work)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                persist_burst()
                e1.record()
                torch.cuda.synchronize()
                bursts.append(e0.elapsed_time(e1) * 1e3 / n_rep)
            ops.check_device_errors()
            t_launch = sorted(bursts[1:])[len(bursts[1:]) // 2]     # us per launch, median of 3 bursts
            us_p = t_launch / n_steps
            w_bytes = 4 * Hh * Hh * 4
            # step 0 has no recurrent product (h = 0): the 23 others are what the 5.77 MB describe;
            # the whole launch (weight load, handshake, step 0) is charged to them
            res_bytes = (step_bytes - w_bytes) * n_steps + w_bytes        # W read once per launch
            lstm = {"bound": "hbm",
                    "kernel": "lstm_persist_kernel (b=64, H=512): %d dependent steps in ONE launch, weights register-resident "
                              "as 2 f16 pieces each (3 v_mfma_f32_16x16x32_f16 products per multiply, f32 accumulate), h "
                              "handed between workgroups through L2" % n_steps,
                    "how": "HIP events on the launch stream around bursts of %d back-to-back launches (median of 3 "
                           "bursts); per step = launch time / %d, so launch gap, weight load, handshake and the "
                           "product-less step 0 are all charged to the steps" % (n_rep, n_steps),
                    "accounting": "SURVEY 8(d): 5.77 MB per step (W 4.19 MB + pre-activations + h,c r/w + gate "
                                  "save), i.e. as if W were re-read every step",
                    "achieved": round(step_bytes / us_p / 1e3, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(step_bytes / us_p / 1e3 / 8000.0, 4),
                    "traffic": pmc_traffic("lstm_persist_bytes_per_step")[0],
                    "traffic_source": pmc_traffic("lstm_persist_bytes_per_step")[1],
                    "bytes_per_step": step_bytes, "us_per_step": round(us_p, 3), "us_per_launch": round(t_launch, 1),
                    "weights_resident": {"accounting": "SURVEY 8(d) weights-resident variant: W counted once per launch",
                                         "bytes_per_step": res_bytes // n_steps,
                                         "achieved": round(res_bytes / t_launch / 1e3, 1),
                                         "frac": round(res_bytes / t_launch / 1e3 / 8000.0, 4)},
                    "launch_per_step": per_launch}

    # the two other rooflines SURVEY.md 8(d) names, timed with events on the launch stream:
    # the vocabulary projection (MFMA) and the attention step after the encoder_att hoist (HBM)
    extra = None
    if rank == 0 and not args.no_lstm_roofline:
        def timed(fn, reps):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps      # us
        Nt, Hh = sum(lengths), 512
        hid = torch.randn(Nt, Hh, device=dev)
        cw = torch.randn(V, Hh, device=dev) * 0.05
        cb = torch.zeros(V, device=dev)
        us_v = timed(lambda: ops.sgemm(hid, cw, transB=True, bias=cb), 20)
        fl = 2.0 * Nt * V * Hh
        bb, P, A, Cf = 64, 196, 512, 2048
        att1 = torch.randn(bb, P, A, device=dev)
        feat = torch.randn(bb, P, Cf, device=dev).abs()
        z = torch.randn(bb, A + Cf, device=dev)
        wf, bf = torch.randn(1, A, device=dev) * 0.1, torch.zeros(1, device=dev)
        us_a = timed(lambda: ops.attention_step(att1, feat, z.clone(), A, wf, bf), 20)
        us_clone = timed(lambda: z.clone(), 20)
        us_a -= us_clone
        by = bb * (P * A * 4 + P * Cf * 4)
        extra = {
            # capnet_sgemm takes gemm_b3.hip for this product unless CAPNET_NO_B3=1: three exact bf16 pieces per fp32 operand,
            # SIX piece products per multiply on the 16-bit matrix pipe -- peak for the ISSUED work = dense bf16 / 6
            "vocab_projection": (lambda b3: {
                "bound": "mfma",
                "kernel": ("gemm_b3_kernel (three bf16 pieces per fp32 operand, six v_mfma_f32_16x16x32_bf16 products per multiply, "
                           "fp32 accumulate; " if b3 else "nt_dma_kernel<128,32> (f32 MFMA, both operands by LDS-DMA; ") +
                          "logits = hiddens . C^T, %d x %d x %d)" % (Nt, V, Hh),
                "achieved": round(fl / us_v / 1e6, 2),
                "peak": round(MFMA_BF16_PEAK_TFLOPS / 6.0, 1) if b3 else MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fl / us_v / 1e6 / (MFMA_BF16_PEAK_TFLOPS / 6.0 if b3 else MFMA_F32_PEAK_TFLOPS), 4),
                "vs_f32_matrix_peak": round(fl / us_v / 1e6 / MFMA_F32_PEAK_TFLOPS, 4), "us": round(us_v, 1)})(
                    os.environ.get("CAPNET_NO_B3") != "1" and float(Nt) * V * Hh >= 2.5e8 and ((Nt + 127) // 128) * (V // 128) >= 192),
            "attention_step": {"bound": "hbm", "kernel": "att_scores_fwd + att_context_fwd (b=64, P=196, A=512, "
                               "C=2048; att1 + feature map read once per row)", "achieved": round(by / us_a / 1e3, 1),
                               "peak": 8000.0, "unit": "GB/s", "frac": round(by / us_a / 1e3 / 8000.0, 4),
                               "bytes_per_step": by, "us_per_step": round(us_a, 1)},
        }

    parity_failed = False
    if rank == 0:
        total_images = B * world * args.steps
        out = {
            "metric": "images/sec (train step) + NLL loss match, batch=64 ResNet152+FactoredLSTM-512",
            "value": round(total_images / elapsed, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (trunk: operands as 2 f16 pieces each, 3 MFMA products per multiply, f32 accumulate; decoder: large products as 3 bf16 pieces per operand, 6 MFMA products per multiply, the rest f32 MFMA; the persistent LSTM kernel the trunk's split-f16 scheme)",
            "data": "synthetic",
            "config": {"workload": "configs[1]: StyleNet FactoredLSTM (factored 512, hidden 512, 1 layer, "
                                   "emb 300, V=%d) + ResNet-152 train-mode trunk, batch %d/GPU, 224x224, "
                                   "tf 0.8, dropout %.2f, clamp 0.5 + Adam 2e-4" % (V, B, args.dropout),
                       "decoder": args.decoder, "global_batch": B * world,
                       "parallelism": "dp%d" % world,
                       "pipeline": ("decoder of step i || trunks of steps i+1..i+%d" % pipe.depth) if pipe is not None else "none"},
            "loss_first": round(float(first.item()), 5) if first is not None else None,
            "loss_last": round(float(last.item()), 5),
            "roofline": roofline,
            "roofline_lstm_step": lstm,
            "roofline_other": extra,
        }
        if args.decoder != "factored":
            out["config"]["workload"] = out["config"]["workload"].replace(
                "configs[1]: StyleNet FactoredLSTM", "secondary (--decoder %s): " % args.decoder)
        if args.layers > 1 or args.factored != 512:
            out["config"]["workload"] = out["config"]["workload"].replace(
                "configs[1]: StyleNet FactoredLSTM (factored 512, hidden 512, 1 layer,",
                "secondary (PERF-ONLY, PARITY UNPINNED: the reference ignores num_layers, capnet.stacked defines the "
                "stacking): FactoredLSTM (factored %d, hidden 512, %d layers," % (args.factored, args.layers))
        if world == 1 and not args.no_cpu_baseline and args.decoder == "factored" and args.layers == 1 and args.factored == 512:
            # `metric` says "+ NLL loss match": the evidence is in the line itself. Untimed.
            parity = gpu_parity_losses(dev, B, V)
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_steps, parity=parity)
            out["loss_parity"] = parity
            if parity["rel_max"] > parity["tolerance"]:
                log("LOSS PARITY FAILED: %s" % parity)
                parity_failed = True
        else:
            out["cpu_baseline"] = None
            out["loss_parity"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("loss parity beyond 1e-4 (the JSON line was still printed)")


if __name__ == "__main__":
    main()
