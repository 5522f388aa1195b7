"""Single-node data parallelism: one process per GPU, RCCL all-reduce of ONE flat gradient
buffer, overlapped with the next step's frozen-trunk forward.

The reference has no multi-GPU code at all; this layer is new. Semantics:
  * rank r trains on its own shard (rows r::G of a globally length-sorted batch keep every
    shard sorted and balance the token counts) with PER-RANK BatchNorm statistics;
  * the reduced gradient is sum_r (N_r / N_global) * grad_r, i.e. the gradient of the
    global token-mean loss (each rank's backward is scaled by N_r/N_global, then a SUM
    all-reduce); every rank then applies the identical fused clamp+Adam update;
  * the teacher-forcing draws must be identical on all ranks: draw them once per step for
    max_r(steps_r) steps (draw_tf_mask below) and pass tf_mask= to the decoder.
Overlap: the all-reduce + update run on a side stream; the compute stream only waits for them
right before the trainable encoder head of the NEXT step (EncoderCNN.pre_head_hook), so they
hide behind that step's ResNet-152 forward, which does not depend on the update.
"""
import random

import os

import torch

from . import ops
from .optim import Adam


def shard_rows(n_rows, rank, world_size):
    """Row indices of `rank`'s shard of a length-sorted global batch."""
    return list(range(rank, n_rows, world_size))


def draw_tf_mask(n_steps, teacher_forcing_ratio):
    """One random.random() draw per step (stylenet/model.py:181); call with the GLOBAL maximum
    number of steps so every rank consumes the same number of draws."""
    return [random.random() < teacher_forcing_ratio for _ in range(n_steps)]


class NativeComm:
    """RCCL through the C ABI (capnet_comm_create / capnet_allreduce_grads, include/capnet.h) instead of through
    torch.distributed: the 128-byte unique id is drawn on rank 0 and handed round with torch.distributed's object
    broadcast (any backend), the communicator binds to the current device, the all-reduce is enqueued on the CURRENT
    torch stream. Opt-in (CAPNET_NATIVE_RCCL=1 or GradAllReducer(native=True)): the default stays torch.distributed,
    whose process group the caller has anyway."""

    def __init__(self, process_group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self._lib, self._C = _lib, C
        if dist.is_initialized():
            rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        else:
            rank, world = 0, 1
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            _lib.check(_lib.lib().capnet_comm_unique_id(buf), "capnet_comm_unique_id")
        ids = [bytes(buf)]
        if world > 1:
            dist.broadcast_object_list(ids, src=0, group=process_group)
        idbuf = (C.c_ubyte * 128).from_buffer_copy(ids[0])
        self.handle = C.c_void_p()
        _lib.check(_lib.lib().capnet_comm_create(idbuf, rank, world, C.byref(self.handle)), "capnet_comm_create")
        self.rank, self.world = rank, world

    def all_reduce(self, flat):
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()):
            raise ops.CapnetError("NativeComm.all_reduce: a contiguous fp32 CUDA tensor is required")
        self._lib.check(self._lib.lib().capnet_allreduce_grads(self.handle, flat.data_ptr(), flat.numel(),
                                                               self._lib.current_stream()), "capnet_allreduce_grads")

    def destroy(self):
        if self.handle:
            self._lib.check(self._lib.lib().capnet_comm_destroy(self.handle), "capnet_comm_destroy")
            self.handle = None


class GradAllReducer:
    """Packs gradients into one flat buffer, all-reduces it, scatters it back.

    pack_fn(tensors, flat) / unpack_fn(tensors, flat, scale) default to the HIP kernels; the
    CPU gloo tests inject torch versions (test infrastructure) to exercise this logic."""

    def __init__(self, process_group=None, pack_fn=None, unpack_fn=None, native=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = process_group
        if native is None:
            native = os.environ.get("CAPNET_NATIVE_RCCL") == "1"
        self.native = NativeComm(process_group) if native else None
        self.pack_fn = pack_fn or (lambda ts, flat: ops.pack_tensors(ts, flat, unpack=False))
        self.unpack_fn = unpack_fn or (lambda ts, flat, s: ops.pack_tensors(ts, flat, unpack=True, scale=s))
        self.flat = None

    def __call__(self, grads, scale=1.0):
        n = sum(g.numel() for g in grads)
        if n == 0:
            return
        if self.flat is None or self.flat.numel() < n + 1 or self.flat.device != grads[0].device:
            self.flat = torch.empty(n + 1, dtype=torch.float32, device=grads[0].device)
        flat = self.flat[:n + 1]
        self.pack_fn(grads, flat[:n])
        # the update kernel drops a step while the LOCAL error word is set: the word rides behind the gradients so that
        # every rank drops the same steps (replicas stay bit-identical)
        share = grads[0].is_cuda
        if share:
            ops.err_word_exchange(flat[n:], 0)
        else:
            flat[n] = 0.0
        if self.native is not None:
            self.native.all_reduce(flat)
        else:
            self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        if share:
            ops.err_word_exchange(flat[n:], 1)
        self.unpack_fn(grads, flat[:n], scale)


class DataParallelAdam(Adam):
    """capnet.optim.Adam whose step() all-reduces the gradients first, on a side stream."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None,
                 overlap=True):
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        import torch.distributed as dist
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # CAPNET_FORCE_ALLREDUCE=1: run the pack / all-reduce / unpack path on a single rank too
        # (rehearsal of the multi-GPU path on a one-GPU box)
        force = dist.is_initialized() and os.environ.get("CAPNET_FORCE_ALLREDUCE") == "1"
        self.reducer = GradAllReducer(process_group) if (self.world_size > 1 or force) else None
        self.overlap = overlap
        self.side = torch.cuda.Stream() if overlap else None
        self.update_done = None

    def attach(self, encoder):
        """Make `encoder` wait for the pending update right before its trainable head."""
        encoder.pre_head_hook = self.wait_for_update
        return self

    def wait_for_update(self):
        if self.update_done is not None:
            torch.cuda.current_stream().wait_event(self.update_done)
            self.update_done = None

    def _reduce_and_update(self):
        if self.reducer is not None:
            grads = []
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is not None:
                        if not p.grad.is_contiguous():
                            p.grad = p.grad.contiguous()
                        grads.append(p.grad)
            self.reducer(grads, 1.0)
        Adam.step(self)

    @torch.no_grad()
    def step(self):
        if not self.overlap:
            self._reduce_and_update()
            return
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        self.side.wait_event(ready)
        with torch.cuda.stream(self.side):
            self._reduce_and_update()
            self.update_done = torch.cuda.Event()
            self.update_done.record(self.side)

    def zero_grad(self, set_to_none=True):
        # gradients are read by the side stream until the update is done
        self.wait_for_update()
        super().zero_grad(set_to_none=set_to_none)
