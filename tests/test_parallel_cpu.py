"""Data-parallel logic on CPU with the gloo backend, world_size 2 (no GPU needed).

The product's pack/unpack are HIP kernels, so these tests inject torch versions (test
infrastructure) into capnet.parallel.GradAllReducer and use the CPU oracle decoder as the model:
what is checked is the sharding, the N_rank/N_global loss weighting + SUM all-reduce (== gradient
of the global token-mean loss) and the shared teacher-forcing draws."""
import os
import random
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import golden_params, load_golden, t
from oracle import decoders_ref as D
from oracle import step_ref as S


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torch_pack(tensors, flat):
    off = 0
    for x in tensors:
        flat[off:off + x.numel()].copy_(x.reshape(-1))
        off += x.numel()


def _torch_unpack(tensors, flat, scale):
    off = 0
    for x in tensors:
        x.copy_((flat[off:off + x.numel()] * scale).view_as(x))
        off += x.numel()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import capnet  # noqa: F401
    from capnet.parallel import GradAllReducer, draw_tf_mask, shard_rows
    torch.set_num_threads(1)
    z = load_golden("decoder_factored_tiny.npz")
    p = golden_params(z)
    captions, lengths, feats = t(z["captions"]), z["lengths"].tolist(), t(z["features"])
    rows = shard_rows(len(lengths), rank, world)
    cap_r, len_r, feat_r = captions[rows], [lengths[i] for i in rows], feats[rows]
    # identical draws on every rank: one per step of the GLOBAL longest sequence
    random.seed(11)
    tf = draw_tf_mask(max(lengths), 0.6)
    n_global = sum(lengths)
    loss, grads, _, _ = S.decoder_loss_and_grads(D.factored_lstm_forward, p, cap_r, len_r, feat_r,
                                                 tf[:max(len_r)], mode="factual")
    names = [k for k, g in grads.items() if g is not None]
    gl = [grads[k] * (float(sum(len_r)) / n_global) for k in names]
    GradAllReducer(pack_fn=_torch_pack, unpack_fn=_torch_unpack)(gl, 1.0)
    if rank == 0:
        torch.save({"names": names, "grads": gl, "tf": tf, "rows": rows}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gives_global_token_mean_gradient(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    assert got["rows"] == [0, 2]
    z = load_golden("decoder_factored_tiny.npz")
    p = golden_params(z)
    captions, lengths, feats = t(z["captions"]), z["lengths"].tolist(), t(z["features"])
    _, ref, _, _ = S.decoder_loss_and_grads(D.factored_lstm_forward, p, captions, lengths, feats,
                                            got["tf"], mode="factual")
    for k, g in zip(got["names"], got["grads"]):
        assert (g - ref[k]).abs().max().item() <= 2e-6 * max(1.0, ref[k].abs().max().item()), k


def test_shards_keep_sorted_lengths_and_balance():
    import capnet  # noqa: F401
    from capnet.parallel import shard_rows
    lengths = sorted([24, 23, 23, 20, 19, 17, 12, 12, 9, 8], reverse=True)
    for world in (2, 4):
        seen = []
        for r in range(world):
            rows = shard_rows(len(lengths), r, world)
            ls = [lengths[i] for i in rows]
            assert ls == sorted(ls, reverse=True)
            seen += rows
        assert sorted(seen) == list(range(len(lengths)))
