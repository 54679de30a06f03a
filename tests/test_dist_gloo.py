"""world_size-2 gloo (CPU) test of the multi-GPU exchange logic in dist_shard.py: row-sharded mu2 gather,
sharded log-sum-exp cross-entropy (partials + combine), their backward exchanges and the flat-arena
gradient all-reduce -- with the CPU oracle as the local compute backend -- against the single-process
full-table result on the global batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as R

C2 = float(1.0 / (2.0 * 0.25))


class OracleBackend:
    """dist_shard backend interface on torch CPU ops (test-only)."""

    def gather_rows(self, shard, idx_all, row0):
        rel = idx_all - row0
        ok = (rel >= 0) & (rel < shard.shape[0])
        out = torch.zeros(idx_all.shape[0], shard.shape[1])
        out[ok] = shard.detach()[rel[ok]]
        return out

    def scatter_rows_(self, dshard, drows, idx_all, row0, scale):
        rel = idx_all - row0
        ok = (rel >= 0) & (rel < dshard.shape[0])
        dshard.index_add_(0, rel[ok], drows[ok] * scale)

    def disc_partials(self, q_all, shard, idx_all, row0):
        lg = R.disc_logits(q_all, shard.detach())
        rmax = lg.max(dim=1).values
        rsum = torch.exp(lg - rmax[:, None]).sum(dim=1)
        rel = idx_all - row0
        ok = (rel >= 0) & (rel < shard.shape[0])
        tgt = torch.zeros_like(rmax)
        tgt[ok] = lg[ok, rel[ok]]
        return torch.stack([rmax, rsum, tgt])

    def pack(self, q, idx):
        return torch.cat([q.detach().float(), idx.to(torch.int32).view(torch.float32).reshape(-1, 1)], dim=1)

    def unpack(self, pk):
        D = pk.shape[1] - 1
        return pk[:, :D].contiguous(), pk[:, D].contiguous().view(torch.int32).to(torch.int64)

    def merge_partials(self, parts):
        m = parts[:, 0].max(dim=0).values
        return m, (parts[:, 1] * torch.exp(parts[:, 0] - m)).sum(dim=0), parts[:, 2].sum(dim=0)

    def bwd_pack(self, dq_all, dq_scale, dmu2_local, own0, n_all, D):
        buf = torch.zeros(n_all, 2 * D)
        if dq_all is not None:
            buf[:, :D] = dq_all * dq_scale
        if dmu2_local is not None:
            buf[own0:own0 + dmu2_local.shape[0], D:] = dmu2_local
        return buf

    def bwd_unpack(self, buf, own0, n_own, want_dq, want_dmu2):
        D = buf.shape[1] // 2
        return (buf[own0:own0 + n_own, :D].contiguous() if want_dq else None), (buf[:, D:].contiguous() if want_dmu2 else None)

    def ce_mean(self, m, s, tgt, scale=1.0):
        return scale * ((m - tgt) + torch.log(s)).mean()

    def disc_bwd(self, q_all, shard, idx_all, row0, m, s, g, g_mul, need_dq, need_dt):
        t = shard.detach()
        lg = R.disc_logits(q_all, t)
        p = torch.exp(lg - m[:, None]) / s[:, None]
        rel = idx_all - row0
        ok = (rel >= 0) & (rel < t.shape[0])
        p[ok, rel[ok]] -= 1.0
        w = p * (g * g_mul)
        diff = q_all[:, None, :] - t[None, :, :]
        dq = (-2 * C2) * (w[:, :, None] * diff).sum(1) if need_dq else None
        dt = (2 * C2) * (w[:, :, None] * diff).sum(0) if need_dt else None
        return dq, dt


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dist_shard as ds

    torch.manual_seed(0)
    S, D, Bl, Fin = 11, 4, 6, 5  # S not divisible by world: ragged last shard
    table = torch.randn(S, D)
    enc = torch.nn.Linear(Fin, D)          # stands in for the z2 encoder (replicated net weights)
    xg = torch.randn(world * Bl, Fin)
    idxg = torch.randint(0, S, (world * Bl,))
    idxg[1] = idxg[Bl]                      # a collision across ranks
    wv = torch.randn(world * Bl, D)
    alpha = 10.0

    # ---- single-process reference on the global batch
    t_ref = table.clone().requires_grad_(True)
    e_ref = torch.nn.Linear(Fin, D)
    e_ref.load_state_dict(enc.state_dict())
    q = e_ref(xg)
    loss_ref = -(R.mu2_gather(t_ref, idxg) * wv * q).sum(1).mean() + alpha * R.disc_loss(q, t_ref, idxg)
    loss_ref.backward()

    # ---- sharded, this rank's slice of the batch
    sh = ds.ShardCtx(S, None, OracleBackend())
    shard = torch.nn.Parameter(table[sh.row0:sh.row1].clone())
    ops = ds.ShardedTableOps(shard, sh)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    ql = enc(xg[sl])
    _, none = ops.lookup(idxg[sl].contiguous(), S)
    assert none is None  # the rows arrive with the CE: ONE packed exchange (all-gather [q | idx], reduce-scatter rows, all-gather partials)
    n_coll = {"n": 0}
    for name in ("all_gather", "all_reduce_", "reduce_scatter"):
        def wrap(fn):
            def f(*a, **k):
                n_coll["n"] += 1
                return fn(*a, **k)
            return f
        setattr(sh, name, wrap(getattr(sh, name)))
    mu2, ce = ops.resolve(ql, None, None, None)
    fwd_coll = n_coll["n"]
    loss = -(mu2 * wv[sl] * ql).sum(1).mean() + alpha * ce
    loss.backward()
    bwd_coll = n_coll["n"] - fwd_coll
    flat = torch.cat([p.grad.reshape(-1) for p in enc.parameters()])
    dist.all_reduce(flat)                   # C1, then 1/W (folded into Adam's grad_scale in the product)
    flat /= world
    ref_flat = torch.cat([p.grad.reshape(-1) for p in e_ref.parameters()])
    ok = {
        "mu2": torch.allclose(mu2.detach(), table[idxg[sl]], atol=1e-6),
        "ce": torch.allclose(ce.detach(), R.disc_loss(e_ref(xg), table, idxg).detach(), rtol=1e-5, atol=1e-6),
        "net_grads": torch.allclose(flat, ref_flat, rtol=1e-4, atol=1e-6),
        "shard_grads": torch.allclose(shard.grad, t_ref.grad[sh.row0:sh.row1], rtol=1e-4, atol=1e-6),
        "rows": (sh.row0, sh.row1),
        # gloo has no reduce-scatter: ShardCtx emulates it with an all-reduce (one nested call) -> 3 (+1) forward, 1 backward
        "collectives": (fwd_coll, bwd_coll),
    }
    ret[rank] = ok
    dist.destroy_process_group()


def test_sharded_table_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0]["rows"] == (0, 6) and ret[1]["rows"] == (6, 11)
    for r in range(world):
        for k, v in ret[r].items():
            if k not in ("rows", "collectives"):
                assert v, "rank %d: %s mismatch" % (r, k)
        # packed exchange: forward = all-gather [q | idx] + reduce-scatter rows (gloo: emulated by an all-reduce inside it, counted
        # twice) + all-gather partials; backward = ONE all-reduce [dq | dmu2]
        assert ret[r]["collectives"] == (4, 1), ret[r]["collectives"]
