"""Adam on the HIP kernel (fhvae_adam_step) with the reference's hyper-parameters
(train_model.py:409-411: Adam(lr, betas=(beta_one, beta_two)), eps 1e-8, no weight decay).

Parameters are packed into ONE flat f32 arena (parameters become views of it) so that the optimizer
is a single elementwise launch per step and data-parallel gradient reduction is one (bucketed)
collective over the flat gradient arena.  The step counter lives on the device so a captured
hipGraph advances the bias correction on replay; the Adam launch counts the step itself and clears the
gradient arena behind its use (FHVAE_ADAM_ADVANCE | FHVAE_ADAM_ZERO_GRAD), so a training step has no
increment launch and no memset: `zero_grad()` right after `step()` finds the arena already zero.
NOTE the one visible difference from torch.optim.Adam: after `step()` every `p.grad` reads zero.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

import hip_binding as hb


class FlatArena:
    """Packs tensors into one flat f32 buffer; `views[i]` aliases the i-th tensor's storage."""

    def __init__(self, tensors: List[torch.Tensor], align: int = 64):
        dev = tensors[0].device
        offs, n = [], 0
        for t in tensors:
            offs.append(n)
            n += (t.numel() + align - 1) // align * align
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets, self.numel = offs, n
        self.views = [self.flat[o:o + t.numel()].view(t.shape) for o, t in zip(offs, tensors)]


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.95, 0.999), eps=1e-8, grad_scale=1.0):
        params = [p for p in params if p.requires_grad]
        if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            raise RuntimeError("FusedAdam needs float32 parameters on a MI355X device")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = float(grad_scale)
        self._params = params
        with torch.no_grad():
            self.p_arena = FlatArena(params)
            self.g_arena = FlatArena(params)
            for p, pv, gv in zip(params, self.p_arena.views, self.g_arena.views):
                pv.copy_(p.data)
                p.data = pv          # the parameter now lives in the arena
                p.grad = gv          # gradients accumulate straight into the flat gradient arena
                p._fh_grad = gv      # ... and the HIP backward kernels write there directly (hip_binding._sink)
        self.m = torch.zeros_like(self.p_arena.flat)
        self.v = torch.zeros_like(self.p_arena.flat)
        self._step_buf = torch.zeros(hb.ADAM_STEP_WORDS, device=params[0].device, dtype=torch.int32)  # [0] the step count, then the kernel's scratch words
        self.step_dev = self._step_buf[0]  # (0-dim view: .item() / .fill_() / .copy_() as before)
        self._zeroed_by_step = False  # the last thing that touched the gradient arena was step(): it is all zeros

    # -- checkpoint contract (utils.py:87,131; train_model.py:415): torch.optim.Adam's state-dict layout, so the moments and the
    # step count survive a save / resume and a state saved by torch.optim.Adam over the same parameters loads here -----------
    def state_dict(self):
        sd = super().state_dict()  # param_groups (+ parameter indices); Optimizer.state itself is empty: the arenas hold it
        step = float(self.step_dev.item())
        state = {}
        for i, (off, p) in enumerate(zip(self.p_arena.offsets, self._params)):
            n = p.numel()
            state[i] = {"step": torch.tensor(step), "exp_avg": self.m[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": self.v[off:off + n].view(p.shape).clone()}
        sd["state"] = state
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._params):
            raise ValueError("FusedAdam.load_state_dict: expected one parameter group of %d parameters" % len(self._params))
        g = self.param_groups[0]
        for k in ("lr", "betas", "eps"):
            if k in groups[0]:
                g[k] = groups[0][k] if k != "betas" else tuple(groups[0][k])
        state = state_dict.get("state", {})
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for j, key in enumerate(groups[0]["params"]):
            st = state.get(key, state.get(str(key)))
            if st is None:
                continue
            off, p = self.p_arena.offsets[j], self._params[j]
            n = p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("FusedAdam.load_state_dict: moment shape %s does not match parameter %s"
                                 % (tuple(st["exp_avg"].shape), tuple(p.shape)))
            self.m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(self.m.device, torch.float32))
            self.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(self.v.device, torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdam keeps ONE step count for all parameters; the state holds %s" % sorted(steps))
        self.step_dev.fill_(steps.pop() if steps else 0)

    def zero_grad(self, set_to_none: bool = False):
        # keep the arena views attached (set_to_none would detach them); one memset for everything
        hb.flush_param_grads()  # (a backward that was never followed by step(): its queued kernels must not land after the memset)
        hb.join_side_stream()
        if self._zeroed_by_step:
            self._zeroed_by_step = False  # (a backward follows: the arena will not be zero the next time)
        else:
            self.g_arena.flat.zero_()
        for p, gv in zip(self._params, self.g_arena.views):
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv

    def flat_grad(self) -> torch.Tensor:
        hb.flush_param_grads()  # the nets' deferred weight-gradient contractions: one grouped launch
        hb.join_side_stream()  # weight-gradient GEMMs may still be running on the side stream
        return self.g_arena.flat

    @torch.no_grad()
    def step(self, closure=None):
        hb.flush_param_grads()
        hb.join_side_stream()
        for p, gv in zip(self._params, self.g_arena.views):
            if p.grad is not None and p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)  # a gradient produced outside the arena (first backward after set_to_none)
                p.grad = gv
        g = self.param_groups[0]
        hb.adam_step_(self.p_arena.flat, self.g_arena.flat, self.m, self.v, self._step_buf, g["lr"], g["betas"][0],
                      g["betas"][1], g["eps"], self.grad_scale, flags=hb.ADAM_ZERO_GRAD | hb.ADAM_ADVANCE)
        self._zeroed_by_step = True
