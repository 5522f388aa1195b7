"""Adam with the reference's element-wise gradient clamp fused into the update kernel.

Replaces `torch.optim.Adam(params, lr=...)` + `clip_gradient(optimizer, grad_clip)` +
`optimizer.step()` of stylenet/train_multitask.py:166-167,388-389 (betas (0.9, 0.999),
eps 1e-8, README.md:28-33). Parameters whose .grad is None are skipped and keep their own
step count, as torch >= 2.0 does after zero_grad(set_to_none=True).
"""
import torch

from . import ops
from ._lib import CapnetError


class Adam:

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        if isinstance(params[0], dict):
            groups = params
        else:
            groups = [{"params": params}]
        self.param_groups = []
        seen = set()
        for g in groups:
            g = dict(g)
            g["params"] = list(g["params"])
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            for p in g["params"]:
                if id(p) in seen:
                    raise ValueError("some parameters appear in more than one parameter group")
                seen.add(id(p))
            self.param_groups.append(g)
        self.state = {}
        self._pending_clip = None
        self._skipped = None           # device counter of steps the update kernel dropped (error word set)
        self._history = []             # per step() since the last error check: the parameters it stepped
        ops.register_optimizer(self)
        self.grad_sync = None  # optional callable(list_of_params) run before the update (data parallel)

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_()
                        p.grad.zero_()

    def state_dict(self):
        """torch.optim.Adam's layout: `state` {index: {step, exp_avg, exp_avg_sq}} and
        `param_groups` with the parameters replaced by their indices in registration order, so the
        reference's checkpoint/resume (stylenet/utils.py:76-90, train_multitask.py:168-177) works
        with this class and the file loads under torch.load(weights_only=True)."""
        index, groups = {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                index.setdefault(id(p), len(index))
                ids.append(index[id(p)])
            packed = {k: v for k, v in g.items() if k != "params"}
            packed["betas"] = tuple(packed["betas"])
            packed["params"] = ids
            groups.append(packed)
        state = {}
        for p, st in self.state.items():
            state[index[id(p)]] = {"step": torch.tensor(float(st["step"])),
                                   "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"]}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        by_index = {}
        for g, saved in zip(self.param_groups, groups):
            if len(saved["params"]) != len(g["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match "
                                 "the size of optimizer's group")
            for k, v in saved.items():
                if k != "params":
                    g[k] = tuple(v) if k == "betas" else v
            for i, p in zip(saved["params"], g["params"]):
                by_index[i] = p
        self.state = {}
        for i, st in state_dict["state"].items():
            p = by_index[int(i)]
            self.state[p] = {
                "step": int(float(st["step"])),
                "exp_avg": st["exp_avg"].to(device=p.device, dtype=p.dtype).clone(),
                "exp_avg_sq": st["exp_avg_sq"].to(device=p.device, dtype=p.dtype).clone()}

    def forget_dropped_steps(self, none_dropped=False):
        """Called by ops.check_device_errors() (which has just synchronised). The update kernel drops a step while the
        device's error word is set; the host had already counted it. Take the dropped steps -- the device counter says
        how many, and they are the LAST ones, the word stays set until the check -- out of the per-parameter step
        counts, so that Adam's bias correction matches the moments on the device. Returns their number."""
        n = 0
        if not none_dropped and self._skipped is not None:
            n = int(self._skipped.item())
            self._skipped.zero_()
            for stepped in self._history[len(self._history) - n:] if n else []:
                for p in stepped:
                    self.state[p]["step"] -= 1
        self._history = []
        return n

    def set_pending_clip(self, grad_clip):
        """Called by utils.clip_gradient: the clamp is applied inside the next step()'s kernel
        (and written back to .grad, so the visible effect equals clamp_ followed by step)."""
        self._pending_clip = float(grad_clip)

    @torch.no_grad()
    def step(self):
        clip = self._pending_clip
        self._pending_clip = None
        if self.grad_sync is not None:
            self.grad_sync([p for g in self.param_groups for p in g["params"] if p.grad is not None])
        stepped = []
        if len(self._history) < 4096:      # loops check the error word every log_step steps; unbounded without checks
            self._history.append(stepped)
        counted = False
        for g in self.param_groups:
            ps, gs, ms, vs, steps = [], [], [], [], []
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise CapnetError("capnet.optim.Adam updates GPU parameters only")
                st = self.state.get(p)
                if st is None:
                    st = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                    self.state[p] = st
                st["step"] += 1
                stepped.append(p)
                if not counted:
                    if self._skipped is None or self._skipped.device != p.device:
                        self._skipped = ops.skip_counter(p.device)
                    ops.count_skipped(self._skipped)
                    counted = True
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                ps.append(p.data)
                gs.append(p.grad)
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
                steps.append(st["step"])
            b1, b2 = g["betas"]
            ops.clamp_adam(ps, gs, ms, vs, steps, g["lr"], b1, b2, g["eps"], clip or 0.0,
                           write_grad=True)
