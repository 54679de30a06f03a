"""Multi-GPU: data parallel over the minibatch + row-sharded mu2 table.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm).  Net weights are
replicated; rank r owns table rows [row0, row1) and their Adam moments.  Per training step:

  forward   all-gather idx, z2_mu (B_local*D*4 bytes per rank)            -- C2 of SURVEY 8e
            rows   : owners gather their rows for ALL global queries, reduce-scatter -> mu2 of the local batch
            K5     : each rank scans ITS rows for all global queries -> (max, sumexp, target) partials,
                     all-reduce(MAX) max, rescale, all-reduce(SUM) (sumexp, target) -> CE (same on all ranks)
  backward  dtable : local rows, complete, no exchange;   dq : reduce-scatter to the query's owner
            dmu2   : all-gather, owners scatter-add
            nets   : ONE all-reduce over the flat gradient arena; 1/W is folded into Adam's grad_scale
All exchanged messages are O(B*D): latency-bound on xGMI, no table rows ever move.

The local arithmetic goes through a `backend` object.  The product backend is `HipBackend` (the HIP kernels);
tests pass an oracle backend to exercise the collective logic on CPU with gloo.  There is no automatic
fallback: `DistributedFHVAE` always uses `HipBackend`.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn


class HipBackend:
    """Local compute on the HIP kernels (include/fhvae_hip.h)."""

    def __init__(self):
        import hip_binding as hb

        hb.load_library()
        self.hb = hb
        self.lp = False  # bf16 compute mode of the model (DistributedFHVAE sets it): K5 on the split-operand bf16 MFMA kernels

    # A rank whose shard is empty (more ranks than rows: row0 == row1 == S) still takes part in every collective with the
    # neutral element of each reduction; the kernels are never launched on zero rows.
    def gather_rows(self, shard, idx_all, row0):
        if shard.shape[0] == 0:
            return torch.zeros(idx_all.shape[0], shard.shape[1], device=shard.device, dtype=torch.float32)
        return self.hb.raw_gather_rows(shard, idx_all, row0)

    def scatter_rows_(self, dshard, drows, idx_all, row0, scale):
        if dshard.shape[0] == 0:
            return
        self.hb.raw_scatter_rows_(dshard, drows, idx_all, row0, scale)

    def disc_partials(self, q_all, shard, idx_all, row0):
        if shard.shape[0] == 0:
            n = q_all.shape[0]
            z = torch.zeros(n, device=q_all.device, dtype=torch.float32)
            return torch.full_like(z, -float("inf")), z, z.clone()
        rmax, rsum, tgt, _ = self.hb.raw_disc_fwd(q_all, shard, idx_all, row0=row0, want_ce=False, lp=self.lp)
        return rmax, rsum, tgt

    def disc_rescale(self, rmax, rsum, m):
        # an empty shard's (-inf, 0) partial rescales to 0 * exp(-inf) = 0 in the kernel: no special case
        return self.hb.raw_disc_rescale(rmax, rsum, m)

    def ce_mean(self, m, s, tgt):
        return self.hb.raw_disc_ce_mean(m, s, tgt)

    def disc_bwd(self, q_all, shard, idx_all, row0, m, s, g, g_mul, need_dq, need_dt):
        if shard.shape[0] == 0:
            return (torch.zeros_like(q_all) if need_dq else None), (torch.zeros_like(shard) if need_dt else None)
        return self.hb.raw_disc_bwd(q_all, shard, idx_all, m, s, g, g_mul, row0=row0, need_dq=need_dq, need_dt=need_dt, lp=self.lp)


class ShardCtx:
    def __init__(self, num_seqs: int, group=None, backend=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.S = int(num_seqs)
        self.per = (self.S + self.world - 1) // self.world
        self.row0 = min(self.S, self.rank * self.per)
        self.row1 = min(self.S, self.row0 + self.per)
        self.backend = backend
        self.is_gloo = dist.get_backend(group) == "gloo"

    # -- collectives over dim 0, equal sizes on every rank --------------------------------------
    # RCCL ("nccl") is the product transport.  With a gloo group (CPU tests; two ranks sharing one GPU in the GPU
    # test) device tensors are staged through the host, which only changes the transport, not the arithmetic.
    def _staged(self, t):
        return self.is_gloo and t.is_cuda

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        if self._staged(x):
            return self.all_gather(x.cpu()).to(x.device)
        out = x.new_empty((self.world * x.shape[0],) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x.contiguous(), group=self.group)
        return out

    def all_reduce_(self, t: torch.Tensor, op=None, async_op=False):
        op = op if op is not None else dist.ReduceOp.SUM
        if self._staged(t):
            c = t.cpu()
            dist.all_reduce(c, op=op, group=self.group)
            t.copy_(c)
            return None
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def broadcast_(self, t: torch.Tensor, src=0):
        if self._staged(t):
            c = t.cpu()
            dist.broadcast(c, src, group=self.group)
            t.copy_(c)
        else:
            dist.broadcast(t, src, group=self.group)

    def reduce_scatter(self, x_all: torch.Tensor) -> torch.Tensor:
        n = x_all.shape[0] // self.world
        if self.is_gloo:  # gloo has no reduce_scatter: all-reduce and keep the own slice
            self.all_reduce_(x_all)
            return x_all[self.rank * n:(self.rank + 1) * n].clone()
        out = x_all.new_empty((n,) + tuple(x_all.shape[1:]))
        dist.reduce_scatter_tensor(out, x_all.contiguous(), group=self.group)
        return out


class _ShardGather(torch.autograd.Function):
    """mu2 rows of the local batch out of the row-sharded table (replaces torch.gather, simple_fhvae.py:53)."""

    @staticmethod
    def forward(ctx, shard, idx_all, sh: ShardCtx):
        rows_all = sh.backend.gather_rows(shard, idx_all, sh.row0)  # zeros for rows owned elsewhere
        ctx.sh, ctx.shape = sh, tuple(shard.shape)
        ctx.sink = getattr(shard, "_fh_grad", None)
        ctx.save_for_backward(idx_all)
        return sh.reduce_scatter(rows_all)

    @staticmethod
    def backward(ctx, dmu2):
        sh = ctx.sh
        (idx_all,) = ctx.saved_tensors
        d_all = sh.all_gather(dmu2.contiguous())
        sink = ctx.sink
        dshard = sink if sink is not None else torch.zeros(ctx.shape, device=dmu2.device, dtype=dmu2.dtype)
        # the objective is the mean over ranks of the local losses: 1/W on every rank's contribution
        sh.backend.scatter_rows_(dshard, d_all, idx_all, sh.row0, 1.0 / sh.world)
        return (None if sink is not None else dshard), None, None


class _ShardDisc(torch.autograd.Function):
    """log-sum-exp cross-entropy of the local queries against the WHOLE (sharded) table; returns the mean
    over the GLOBAL batch (identical on every rank) -- simple_fhvae.py:119-122."""

    @staticmethod
    def forward(ctx, q_local, shard, idx_all, sh: ShardCtx):
        be = sh.backend
        q_all = sh.all_gather(q_local.detach())
        rmax, rsum, tgt = be.disc_partials(q_all, shard, idx_all, sh.row0)
        m = rmax.clone()
        sh.all_reduce_(m, op=dist.ReduceOp.MAX)
        st = torch.stack([be.disc_rescale(rmax, rsum, m), tgt])
        sh.all_reduce_(st)
        ce = be.ce_mean(m, st[0].contiguous(), st[1].contiguous())
        ctx.sh = sh
        ctx.save_for_backward(q_all, shard, idx_all, m, st[0].contiguous())
        return ce

    @staticmethod
    def backward(ctx, g):
        sh, be = ctx.sh, ctx.sh.backend
        q_all, shard, idx_all, m, s = ctx.saved_tensors
        g = g.reshape(1).contiguous()
        n_all = q_all.shape[0]
        n_loc = n_all // sh.world
        dq_local = dshard = None
        if ctx.needs_input_grad[0]:
            # net gradients are averaged over ranks afterwards, so the query side carries W/B_global = 1/B_local
            dq_all, _ = be.disc_bwd(q_all, shard, idx_all, sh.row0, m, s, g, 1.0 / n_loc, True, False)
            dq_local = sh.reduce_scatter(dq_all)
        if ctx.needs_input_grad[1]:
            # the shard's gradient is complete locally (all global queries were scanned): true scale 1/B_global
            _, dshard = be.disc_bwd(q_all, shard, idx_all, sh.row0, m, s, g, 1.0 / n_all, False, True)
        return dq_local, dshard, None, None


class ShardedTableOps:
    """Plugs into FHVAEBase (`model.table_ops`): same two calls as the single-GPU ops."""

    def __init__(self, shard: nn.Parameter, sh: ShardCtx):
        self.shard, self.sh = shard, sh
        self._idx_all = None

    def lookup(self, mu_idx, num_seqs, mu2_table=None):
        if mu2_table is not None:
            raise ValueError("mu2_table injection is a single-GPU parity feature")
        if int(num_seqs) != self.sh.S:
            raise ValueError("num_seqs=%d does not match the sharded table (%d rows)" % (num_seqs, self.sh.S))
        self._idx_all = self.sh.all_gather(mu_idx)
        return self.shard, _ShardGather.apply(self.shard, self._idx_all, self.sh)

    def disc(self, z2_mu, table, mu_idx):
        return _ShardDisc.apply(z2_mu, self.shard, self._idx_all, self.sh)


class DistributedFHVAE:
    """Wraps a drop-in model (fhvae.FHVAE / simple_fhvae.SimpleFHVAE built with num_seqs=S on this rank's GPU)
    for one-process-per-GPU training.  Every rank must construct the model with the same seed."""

    def __init__(self, model, lr=1e-3, betas=(0.95, 0.999), eps=1e-8, group=None):
        from hip_optim import FusedAdam

        if model.mu2_table is None:
            raise ValueError("build the model with num_seqs= so the table exists")
        self.model = model
        self.sh = ShardCtx(model.mu2_table.shape[0], group, HipBackend())
        self.sh.backend.lp = getattr(model, "compute_dtype", "f32") == "bf16"
        full = model.mu2_table.data
        self.shard = nn.Parameter(full[self.sh.row0:self.sh.row1].clone())
        model.mu2_table = None  # the full table is dropped: only the shard stays resident
        model.table_ops = ShardedTableOps(self.shard, self.sh)
        # Gradient buckets in the order their backward COMPLETES (decoder first, z2 encoder last): each bucket is
        # contiguous in the flat arena, and its all-reduce is started (async, RCCL's own stream) as soon as the net's
        # backward recurrence has been enqueued, i.e. it overlaps the next net's latency-bound backward cells (C1).
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]

        def group_of(name):
            if name.startswith(("pre_decoder", "dec_gauss_layer")):
                return 0
            return 1 if name.startswith("z1_") else 2

        nets = [p for g in (0, 1, 2) for n, p in named if group_of(n) == g]
        counts = [sum(1 for n, _ in named if group_of(n) == g) for g in (0, 1, 2)]
        for p in nets:  # replicas start identical
            self.sh.broadcast_(p.data, 0)
        self.opt_nets = FusedAdam(nets, lr=lr, betas=betas, eps=eps, grad_scale=1.0 / self.sh.world)
        self.opt_table = FusedAdam([self.shard], lr=lr, betas=betas, eps=eps, grad_scale=1.0)
        offs = self.opt_nets.g_arena.offsets + [self.opt_nets.g_arena.numel]
        bounds, k = [], 0
        for c in counts:
            bounds.append((offs[k], offs[k + c]))
            k += c
        self._buckets = bounds                      # [(begin, end)) of each group in the flat gradient arena
        self._bucket_of_ptr = {}
        k = 0
        for g, c in enumerate(counts):
            for v in self.opt_nets.g_arena.views[k:k + c]:
                self._bucket_of_ptr[v.data_ptr()] = g
            k += c
        self._pending = {}
        self.overlap = True

    def _on_lstm_bwd_done(self, sinks):
        """Fired by hip_binding at the end of an LSTM net's backward: that net's Gaussian head ran earlier in the
        backward pass, so the whole bucket is final -> start its all-reduce now."""
        import hip_binding as hb

        # The persistent LSTM kernels need all 256 CUs co-resident: a collective kernel spinning on its peers next to
        # them would hold some of those CUs for as long as the slowest rank takes.  With that schedule the buckets of the
        # first two nets are reduced under the LAST net's weight-gradient contractions instead (_on_lstm_rec_done), where
        # no persistent kernel follows before the optimizer.
        if not self.overlap or not sinks or sinks[0] is None or hb.LAST_LSTM_FORM["form"] != 0:
            return
        g = self._bucket_of_ptr.get(sinks[0].data_ptr())
        if g is None or g in self._pending:
            return
        b, e = self._buckets[g]
        h = self.sh.all_reduce_(self.opt_nets.g_arena.flat[b:e], async_op=True)
        self._pending[g] = h  # None when the transport was synchronous (staged gloo)

    def _on_lstm_rec_done(self, sinks):
        """Fired by hip_binding between a net's backward recurrence and its parameter-gradient contractions.  Persistent
        schedule only: once the LAST net's recurrence is enqueued, the finished buckets of the earlier nets (contiguous in
        the arena) go out as one async all-reduce that overlaps this net's weight gradients and its head's backward."""
        import hip_binding as hb

        if not self.overlap or not sinks or sinks[0] is None or hb.LAST_LSTM_FORM["form"] == 0:
            return
        g = self._bucket_of_ptr.get(sinks[0].data_ptr())
        last = len(self._buckets) - 1
        if g != last or self._pending or last < 1:
            return
        b, e = self._buckets[0][0], self._buckets[last - 1][1]
        if e > b:
            h = self.sh.all_reduce_(self.opt_nets.g_arena.flat[b:e], async_op=True)
            for k in range(last):
                self._pending[k] = h if k == 0 else None

    def _reduce_gradients(self):
        flat = self.opt_nets.flat_grad()
        if not self._pending and self._buckets:  # nothing in flight: one collective over the contiguous buckets
            b, e = self._buckets[0][0], self._buckets[-1][1]
            if e > b:
                self.sh.all_reduce_(flat[b:e])
            return
        for g, (b, e) in enumerate(self._buckets):
            if g in self._pending:
                if self._pending[g] is not None:
                    self._pending[g].wait()
            elif e > b:
                self.sh.all_reduce_(flat[b:e])
        self._pending.clear()

    # -- checkpointing: the full table (and its Adam moments) exists only as row shards ----------------------------------
    def _gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """Concatenate per-rank row blocks (rank order = row order) on every rank; shards are padded to `per` rows for the
        equal-size collective and the padding is cut off again."""
        sh = self.sh
        pad = local.new_zeros((sh.per,) + tuple(local.shape[1:]))
        pad[: local.shape[0]] = local
        return sh.all_gather(pad)[: sh.S]

    def gather_table(self) -> torch.Tensor:
        """The whole (S, D) mu2 table on every rank (north_star's all-gather of mu2 shards: checkpoint / evaluation only,
        never on the training step)."""
        return self._gather_rows(self.shard.data)

    def _named_net_params(self):
        return [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]

    def state_dict(self) -> dict:
        """Single-GPU-compatible state: the model's state_dict with the gathered `mu2_table`, and one Adam state in
        torch.optim.Adam's layout over the parameters in the order model.named_parameters() has on ONE GPU (the module's own
        `mu2_table` first, then the nets) -- what FusedAdam(model.parameters()).state_dict() holds there.
        Collective: call on every rank."""
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        sd["mu2_table"] = self.gather_table()
        nets = self.opt_nets.state_dict()
        tab = self.opt_table.state_dict()
        named = self._named_net_params()
        slot = {id(p): i for i, p in enumerate(self.opt_nets._params)}
        t = tab["state"][0]
        state = {0: {"step": t["step"], "exp_avg": self._gather_rows(t["exp_avg"]), "exp_avg_sq": self._gather_rows(t["exp_avg_sq"])}}
        for j, (n, p) in enumerate(named):
            state[j + 1] = nets["state"][slot[id(p)]]
        groups = [dict(nets["param_groups"][0], params=list(range(len(named) + 1)))]
        return {"state_dict": sd, "optimizer": {"state": state, "param_groups": groups},
                "param_names": ["mu2_table"] + [n for n, _ in named]}

    def load_state_dict(self, full: dict):
        """Inverse of state_dict(): every rank passes the same full state and keeps its own rows."""
        sh = self.sh
        sd = dict(full["state_dict"])
        table = sd.pop("mu2_table")
        self.model.load_state_dict(sd, strict=False)
        with torch.no_grad():
            self.shard.data.copy_(table[sh.row0:sh.row1].to(self.shard.device))
        opt = full.get("optimizer")
        if opt is not None:
            named = self._named_net_params()
            pos = {id(p): j + 1 for j, (_, p) in enumerate(named)}
            st = opt["state"]
            nets_state = {i: st[pos[id(p)]] for i, p in enumerate(self.opt_nets._params) if pos[id(p)] in st}
            g = dict(opt["param_groups"][0])
            self.opt_nets.load_state_dict({"state": nets_state, "param_groups": [dict(g, params=list(range(len(self.opt_nets._params))))]})
            t = st.get(0)
            if t is not None:
                rows = {"step": t["step"], "exp_avg": t["exp_avg"][sh.row0:sh.row1], "exp_avg_sq": t["exp_avg_sq"][sh.row0:sh.row1]}
                self.opt_table.load_state_dict({"state": {0: rows}, "param_groups": [dict(g, params=[0])]})

    def check_status(self) -> int:
        """One host sync (call once per epoch / before a checkpoint, not per step): 0 = healthy; 2 = a NaN lower bound was seen
        (train_model.py:464-466); 3 = a persistent LSTM recurrence launch gave up (e.g. a collective kernel held CUs while it
        ran): results since are invalid.  The worst code over all ranks is returned on every rank."""
        import hip_binding as hb

        code = 2 if hb.diverged(self.shard.device) else (3 if hb.lstm_sync_status() != 0 else 0)
        t = torch.tensor([code], device=self.shard.device, dtype=torch.int32)
        self.sh.all_reduce_(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    def train_step(self, x, idx, nsegs, alpha=10.0):
        from train_model import loss_function

        import hip_binding as hb

        self.opt_nets.zero_grad()
        self.opt_table.zero_grad()
        out = self.model(x, idx, self.sh.S, nsegs)
        loss = loss_function(out[0], out[1], alpha)
        hb.LSTM_BWD_DONE_HOOK["fn"] = self._on_lstm_bwd_done
        hb.LSTM_BWD_REC_HOOK["fn"] = self._on_lstm_rec_done
        try:
            loss.backward()
        finally:
            hb.LSTM_BWD_DONE_HOOK["fn"] = None
            hb.LSTM_BWD_REC_HOOK["fn"] = None
        self._reduce_gradients()  # C1: three bucket all-reduces, two of them already in flight under the backward
        self.opt_nets.step()
        self.opt_table.step()
        return loss.detach(), out[0].detach()
