"""Multi-GPU: data parallel over the minibatch + row-sharded mu2 table.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm).  Net weights are
replicated; rank r owns table rows [row0, row1) and their Adam moments.  FIVE collectives per training step,
all latency-bound (O(B*D) bytes; no table row ever moves):

  forward   (1) all-gather  [z2_mu | idx]            one packed (B_local, D+1) f32 buffer (idx as int32 bits)   -- C2 of SURVEY 8e
            (2) reduce-scatter of the owners' rows   -> mu2 of the local batch (owners gather their rows for ALL global queries)
            (3) all-gather of (max, sumexp, target)  each rank scans ITS rows for all global queries (K5); the W partials are
                                                     combined locally (log-sum-exp merge): replaces all-reduce(MAX) + all-reduce(SUM)
  backward  (4) all-reduce  [dq | dmu2]              one (B_global, 2D) buffer: every rank's partial dq for all queries (summed,
                                                     the owner keeps its slice) and the local queries' dmu2 in their rows (the row
                                                     owners scatter-add them); dtable of the CE term is complete locally
            (5) all-reduce of the net gradients      flat arena; 1/W is folded into Adam's grad_scale; the decoder + z1 buckets go out
                                                     under the z2 encoder's weight-gradient launch (DistributedFHVAE._on_lstm_rec_done)

The local arithmetic goes through a `backend` object.  The product backend is `HipBackend` (the HIP kernels);
tests pass an oracle backend to exercise the collective logic on CPU with gloo.  There is no automatic
fallback: `DistributedFHVAE` always uses `HipBackend`.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

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
        """(3, N): row_max, row_sumexp, tgt_logit of this shard for all queries."""
        n = q_all.shape[0]
        out = torch.empty(3, n, device=q_all.device, dtype=torch.float32)
        if shard.shape[0] == 0:
            out[0].fill_(-float("inf"))
            out[1:].zero_()
            return out
        self.hb.raw_disc_fwd(q_all, shard, idx_all, row0=row0, want_ce=False, lp=self.lp, out3=out)
        return out

    # the small elementwise steps of the exchange, one launch each (fhvae_shard_* / fhvae_disc_merge_partials)
    def pack(self, q, idx):
        return self.hb.shard_pack(q.detach().float().contiguous(), idx.contiguous())

    def unpack(self, pk):
        return self.hb.shard_unpack(pk)

    def merge_partials(self, parts):
        return self.hb.disc_merge_partials(parts)

    def bwd_pack(self, dq_all, dq_scale, dmu2_local, own0, n_all, D):
        return self.hb.shard_bwd_pack(dq_all, dq_scale, dmu2_local.contiguous() if dmu2_local is not None else None, own0, n_all, D)

    def bwd_unpack(self, buf, own0, n_own, want_dq, want_dmu2):
        return self.hb.shard_bwd_unpack(buf, own0, n_own, want_dq, want_dmu2)

    def ce_mean(self, m, s, tgt, scale=1.0):
        return self.hb.raw_disc_ce_mean(m, s, tgt, scale)

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
        # one rank: the collectives are identities and are skipped (FHVAE_DIST_NO_SOLO=1 issues them anyway: the tests' way to
        # drive the RCCL calls -- and their capture into a hipGraph -- on a one-GPU box)
        self.solo = self.world == 1 and not os.environ.get("FHVAE_DIST_NO_SOLO")

    # -- collectives over dim 0, equal sizes on every rank --------------------------------------
    # RCCL ("nccl") is the product transport.  With a gloo group (CPU tests; two ranks sharing one GPU in the GPU
    # test) device tensors are staged through the host, which only changes the transport, not the arithmetic.
    def _staged(self, t):
        return self.is_gloo and t.is_cuda

    # A one-rank group moves nothing: every collective is the identity and is not issued (RCCL would run it as ~8 device
    # memcpy / memset launches each: 0.16 ms per step at the bench shape).
    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        if self.solo:
            return x.contiguous()
        if self._staged(x):
            return self.all_gather(x.cpu()).to(x.device)
        out = x.new_empty((self.world * x.shape[0],) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x.contiguous(), group=self.group)
        return out

    def all_reduce_(self, t: torch.Tensor, op=None, async_op=False):
        op = op if op is not None else dist.ReduceOp.SUM
        if self.solo:
            return None
        if self._staged(t):
            c = t.cpu()
            dist.all_reduce(c, op=op, group=self.group)
            t.copy_(c)
            return None
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def broadcast_(self, t: torch.Tensor, src=0):
        if self.solo:
            return
        if self._staged(t):
            c = t.cpu()
            dist.broadcast(c, src, group=self.group)
            t.copy_(c)
        else:
            dist.broadcast(t, src, group=self.group)

    def reduce_scatter(self, x_all: torch.Tensor) -> torch.Tensor:
        n = x_all.shape[0] // self.world
        if self.solo:
            return x_all
        if self.is_gloo:  # gloo has no reduce_scatter: all-reduce and keep the own slice
            self.all_reduce_(x_all)
            return x_all[self.rank * n:(self.rank + 1) * n].clone()
        out = x_all.new_empty((n,) + tuple(x_all.shape[1:]))
        dist.reduce_scatter_tensor(out, x_all.contiguous(), group=self.group)
        return out


class _ShardTable(torch.autograd.Function):
    """(mu2 rows of the local batch, CE of the local queries against the WHOLE table) out of the row-sharded table: replaces
    torch.gather (simple_fhvae.py:53) and the (B,S,D) log-sum-exp cross-entropy (simple_fhvae.py:119-122; the mean over the
    GLOBAL batch, identical on every rank).  Three collectives forward, one backward (module docstring)."""

    @staticmethod
    def forward(ctx, q_local, shard, idx_local, sh: ShardCtx, sign=1.0):
        be = sh.backend
        ctx.sign = float(sign)
        # (1) one buffer: the queries and, in the last column, the row indices as int32 bit patterns
        if sh.solo:  # one rank: the exchange is the identity -- no pack / unpack / merge launches either (like the collectives)
            q_all, idx_all = q_local.detach().float().contiguous(), idx_local.contiguous()
        else:
            q_all, idx_all = be.unpack(sh.all_gather(be.pack(q_local, idx_local)))
        # (2) rows: zeros for rows owned elsewhere, so the sum over ranks is the row
        mu2 = sh.reduce_scatter(be.gather_rows(shard, idx_all, sh.row0))
        # (3) K5 partials of this shard for ALL queries, merged locally (an empty shard's (-inf, 0) contributes nothing)
        parts = sh.all_gather(be.disc_partials(q_all, shard, idx_all, sh.row0)).view(sh.world, 3, -1)
        m, s, t = (parts[0, 0], parts[0, 1], parts[0, 2]) if sh.solo else be.merge_partials(parts)
        ce = be.ce_mean(m, s, t, ctx.sign)  # sign * CE (fhvae_core.FHVAEBase._tail)
        ctx.sh, ctx.shape = sh, tuple(shard.shape)
        ctx.sink = getattr(shard, "_fh_grad", None)
        ctx.save_for_backward(q_all, shard, idx_all, m, s)
        ctx.mark_non_differentiable(idx_all)
        return mu2, ce, idx_all

    @staticmethod
    def backward(ctx, dmu2, g, _):
        sh, be = ctx.sh, ctx.sh.backend
        q_all, shard, idx_all, m, s = ctx.saved_tensors
        n_all, D = q_all.shape
        n_loc = n_all // sh.world
        need_dq, need_dt = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dq_all = dshard = None
        if g is not None and (need_dq or need_dt):
            # one call for both sides (true scale 1/B_global: the shard's gradient is complete locally, all global queries were
            # scanned); the query side is averaged over ranks with the net gradients afterwards, so it carries W/B_global
            dq_all, dshard = be.disc_bwd(q_all, shard, idx_all, sh.row0, m, s, g.reshape(1).contiguous(), ctx.sign / n_all, need_dq, need_dt)
        # (4) [dq for all queries | dmu2 of the local queries in their rows], summed over the ranks
        if sh.solo:  # one rank: both halves are already what the unpack would return
            dq_local = dq_all if need_dq else None
            dmu2_all = dmu2.contiguous() if (need_dt and dmu2 is not None) else (torch.zeros_like(q_all) if need_dt else None)
        else:
            buf = be.bwd_pack(dq_all, float(sh.world), dmu2, sh.rank * n_loc, n_all, D)
            sh.all_reduce_(buf)
            dq_local, dmu2_all = be.bwd_unpack(buf, sh.rank * n_loc, n_loc, need_dq, need_dt)
        if need_dt:
            sink = ctx.sink
            if sink is None:
                sink = dshard if dshard is not None else q_all.new_zeros(ctx.shape)
                dshard = sink
            # the objective is the mean over ranks of the local losses: 1/W on every rank's mu2 contribution
            be.scatter_rows_(sink, dmu2_all, idx_all, sh.row0, 1.0 / sh.world)
        return dq_local, dshard, None, None, None


class ShardedTableOps:
    """Plugs into FHVAEBase (`model.table_ops`): same calls as the single-GPU ops; the exchange happens in `resolve`."""

    def __init__(self, shard: nn.Parameter, sh: ShardCtx):
        self.shard, self.sh = shard, sh
        self._idx = None

    def lookup(self, mu_idx, num_seqs, mu2_table=None):
        if mu2_table is not None:
            raise ValueError("mu2_table injection is a single-GPU parity feature")
        if int(num_seqs) != self.sh.S:
            raise ValueError("num_seqs=%d does not match the sharded table (%d rows)" % (num_seqs, self.sh.S))
        self._idx = mu_idx
        return self.shard, None  # the rows arrive with the CE in resolve(): one exchange for both

    def resolve(self, z2_mu, table, mu_idx, mu2, sign=1.0):
        mu2, ce, _ = _ShardTable.apply(z2_mu, self.shard, self._idx if mu_idx is None else mu_idx, self.sh, float(sign))
        return mu2, ce


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
        self.sh.backend.hb.reset_device_words(model.mu2_table.device)  # the sticky status words are per process: start clean
        full = model.mu2_table.data
        self.shard = nn.Parameter(full[self.sh.row0:self.sh.row1].clone())
        model.mu2_table = None  # the full table is dropped: only the shard stays resident
        model.table_ops = ShardedTableOps(self.shard, self.sh)
        # Gradient buckets in the order their backward COMPLETES (decoder first, z2 encoder last): each bucket is contiguous
        # in the flat arena; the first ones are reduced under the last net's weight-gradient launch (_on_lstm_rec_done).
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
        self._pending = None  # handle of the early all-reduce over the first buckets (False: it was synchronous)
        # one rank: the all-reduces move nothing, so there is nothing to hide behind the last net's weight gradients and the
        # grouped weight-gradient launch stays whole (split in two it costs ~60 us at the bench shape)
        self.overlap = self.sh.world > 1

    def _on_lstm_rec_done(self, sinks):
        """Fired by hip_binding when a net's backward RECURRENCE has been enqueued (with deferred parameter gradients: queued
        behind it).  Nets finish in bucket order.  When the second-to-last net (the z1 encoder) is through, the queued weight
        gradients so far (decoder + z1 encoder) are flushed as one grouped launch; when the LAST net's recurrence is enqueued,
        those buckets are complete and go out as ONE async all-reduce.  RCCL's stream waits for the work enqueued so far, i.e.
        for the last persistent recurrence kernel of the step -- those need all 256 CUs co-resident, a collective kernel
        spinning on its peers must not run beside them -- and the reduction then overlaps the last net's weight-gradient
        launch (the second grouped launch, flushed by flat_grad()) and its head's backward."""
        import hip_binding as hb

        if not self.overlap or not sinks or sinks[0] is None or self._pending is not None:
            return
        g = self._bucket_of_ptr.get(sinks[0].data_ptr())
        last = len(self._buckets) - 1
        if g is None or last < 1:
            return
        if g == last - 1:
            hb.flush_param_grads()
        elif g == last:
            hb.flush_param_grads_except_last()  # (no-op unless a net in between never fired: keeps the invariant explicit)
            b, e = self._buckets[0][0], self._buckets[last - 1][1]
            if e > b:
                h = self.sh.all_reduce_(self.opt_nets.g_arena.flat[b:e], async_op=True)
                self._pending = h if h is not None else False  # None: the transport was synchronous (staged gloo)

    def _drop_stale_pending(self):
        """An early all-reduce left over from a step whose backward raised after it was issued: wait it out and forget it, or the
        next step would skip its own early all-reduce (the hook returns while a handle is pending) and reduce only the last
        bucket -- the ranks would silently diverge."""
        if self._pending not in (None, False):
            self._pending.wait()
        self._pending = None

    def _reduce_gradients(self):
        flat = self.opt_nets.flat_grad()  # flushes the weight gradients still queued (the last net's: the second grouped launch)
        if not self._buckets:
            return
        if self._pending is None:  # nothing went out early (FC model, overlap off): one collective over all buckets
            b, e = self._buckets[0][0], self._buckets[-1][1]
            if e > b:
                self.sh.all_reduce_(flat[b:e])
            return
        b, e = self._buckets[-1]
        if e > b:
            self.sh.all_reduce_(flat[b:e])
        if self._pending is not False:
            self._pending.wait()
        self._pending = None

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

        self._drop_stale_pending()
        self.opt_nets.zero_grad()
        self.opt_table.zero_grad()
        out = self.model(x, idx, self.sh.S, nsegs)
        loss = loss_function(out[0], out[1], alpha)
        hb.LSTM_BWD_REC_HOOK["fn"] = self._on_lstm_rec_done
        try:
            hb.backward(loss)
        finally:
            hb.LSTM_BWD_REC_HOOK["fn"] = None
        self._reduce_gradients()  # C1: the decoder + z1 buckets are already in flight under the z2 encoder's weight gradients
        self.opt_nets.step()
        self.opt_table.step()
        return loss.detach(), out[0].detach()
