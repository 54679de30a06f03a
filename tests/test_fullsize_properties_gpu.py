"""Parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle would need minutes and
gigabytes there): translation invariance and softmax normalisation of the discriminative loss, shard-combine
associativity, additivity of the bound, linearity of the LSTM backward, gather/scatter adjointness."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_ops_gpu import close, dev, hb  # noqa: F401


@pytest.mark.parametrize("B,S", [(2048, 28000), (2048, 125000)])  # configs[2]; one GPU's shard of configs[4] (1M / 8)
def test_disc_translation_invariance_and_shard_associativity(hb, B, S):
    D = 32
    g = torch.Generator().manual_seed(B + S)
    q = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
    t = torch.randn(S, D, generator=g).cuda().requires_grad_(True)
    idx = torch.randint(0, S, (B,), generator=g).cuda()
    ce = hb.disc_lse(q, t, idx)
    ce.backward()
    # logits depend on q - t only: the gradient w.r.t. a common translation vanishes
    resid = q.grad.sum(0) + t.grad.sum(0)
    scale = q.grad.abs().sum(0)
    assert (resid.abs() <= 2e-4 * scale + 1e-7).all(), (resid.abs().max().item(), scale.min().item())
    # a translated problem has the same loss
    shift = torch.randn(D, generator=g).cuda() * 0.5
    ce2 = hb.disc_lse(q.detach() + shift, t.detach() + shift, idx)
    close(ce2, ce.detach(), rtol=1e-4, what="translated CE")
    # three row shards combine to the unsharded (max, sumexp, target)
    rmax, rsum, tgt, _ = hb.raw_disc_fwd(q.detach(), t.detach(), idx, want_ce=False)
    cuts = [0, S // 3, S // 3 + 1, S]
    parts = torch.empty(len(cuts) - 1, 3, q.shape[0], device="cuda")  # what the ranks all-gather (dist_shard._ShardTable)
    for w, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        hb.raw_disc_fwd(q.detach(), t.detach()[a:b].contiguous(), idx, row0=a, want_ce=False, out3=parts[w])
    m, s, tg = hb.disc_merge_partials(parts)
    close(m, rmax, rtol=1e-6, what="combined max")
    close(torch.log(s) + m, torch.log(rsum) + rmax, rtol=1e-5, what="combined lse")
    close(tg, tgt, rtol=1e-5, what="combined target logit")
    close(hb.raw_disc_ce_mean(m, s, tg), ce.detach(), rtol=1e-5, what="combined CE")
    # the exchange's pack / unpack round trips (indices as int32 bit patterns beside the f32 queries)
    q2, i2 = hb.shard_unpack(hb.shard_pack(q.detach(), idx))
    assert torch.equal(q2, q.detach()) and torch.equal(i2, idx)
    dq, dm = torch.randn(q.shape[0], q.shape[1], device="cuda"), torch.randn(100, q.shape[1], device="cuda")
    buf = hb.shard_bwd_pack(dq, 2.0, dm, 50, q.shape[0], q.shape[1])
    dql, dma = hb.shard_bwd_unpack(buf, 50, 100)
    assert torch.equal(dql, dq[50:150] * 2.0) and torch.equal(dma[50:150], dm) and float(dma[:50].abs().sum() + dma[150:].abs().sum()) == 0.0


def test_elbo_additivity_fullsize(hb):
    B, T, F, D = 2048, 20, 80, 32
    g = torch.Generator().manual_seed(3)
    x, xm = torch.randn(T, B, F, generator=g).cuda(), torch.randn(T, B, F, generator=g).cuda()
    xl = (torch.randn(T, B, F, generator=g) * 0.4).cuda()
    z = [torch.randn(B, D, generator=g).cuda() for _ in range(5)]
    ns = torch.randint(20, 200, (B,), generator=g).cuda()
    lb, lpx, k1, k2, pm = hb.elbo(x, xm, xl, *z, ns, (B, T, F, (F, B * F), (F, B * F)), True)
    close(lb, lpx + k1 + k2 + pm / ns, rtol=1e-6, what="lower_bound = sum of its terms (simple_fhvae.py:116)")
    assert (k1 <= 1e-3).all() and (k2 <= 1e-3).all() and (pm < 0).all()  # -KL <= 0, log N(.;0,1) < 0


def test_lstm_backward_is_linear_in_the_upstream_gradient_fullsize(hb):
    B, T, I, H, L = 256, 20, 80, 256, 2
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(I, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    x = torch.randn(T, B, I).cuda()
    g1, g2 = torch.randn(T, B, H).cuda(), torch.randn(B, L * H).cuda()

    def grads(a, b):
        params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
        hs, hn = hb.lstm_seq(x, None, T, params, hb.F32)
        ((hs * a).sum() + (hn * b).sum()).backward()
        return [p.grad for p in params], hs.detach()

    ga, hs_a = grads(g1, g2)
    gb, hs_b = grads(2.0 * g1, 2.0 * g2)
    assert torch.equal(hs_a, hs_b)  # the forward is deterministic
    for a, b, n in zip(ga, gb, names):
        close(b, 2.0 * a, rtol=2e-4, what="linearity " + n)  # split-K atomics reorder the f32 sums


def test_gather_scatter_adjoint_fullsize(hb):
    S, D, B = 1_000_000, 32, 2048
    g = torch.Generator().manual_seed(5)
    table = torch.randn(S, D, generator=g).cuda()
    idx = torch.randint(0, S, (B,), generator=g).cuda()
    u = torch.randn(B, D, generator=g).cuda()
    rows = hb.raw_gather_rows(table, idx)
    assert torch.equal(rows, table[idx])
    dt = torch.zeros(S, D, device="cuda")
    hb.raw_scatter_rows_(dt, u, idx)
    # <gather(table), u> == <table, scatter(u)>
    close((rows * u).sum(), (table * dt).sum(), rtol=1e-4, what="adjoint")


@pytest.mark.parametrize("B", [256, 2048])
def test_bf16_lstm_rows_are_independent_fullsize(hb, B):
    """configs[1] shape (2x256 LSTM, T=20, F=80), bf16 persistent kernels: a segment's outputs and its contribution to the
    gradients do not depend on where it sits in the batch -- permuting the batch rows permutes hs/hn BIT FOR BIT (rows only
    meet in the weight gradients), and the weight gradients of the permuted batch agree to split-K summation order.  The
    persistent kernels cut the batch into clusters / row tiles / waves: any cross-row leak or stale exchange shows here."""
    T, I, H, L = 20, 80, 256, 2
    torch.manual_seed(11)
    lstm = torch.nn.LSTM(I, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    x = torch.randn(T, B, I).cuda()
    g1, g2 = torch.randn(T, B, H).cuda(), torch.randn(B, L * H).cuda()
    perm = torch.randperm(B).cuda()

    def run(xx, a, b):
        params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
        hs, hn = hb.lstm_seq(xx, None, T, params, hb.BF16)
        ((hs * a).sum() + (hn * b).sum()).backward()
        torch.cuda.synchronize()
        return hs.detach(), hn.detach(), [p.grad for p in params]

    hs, hn, gr = run(x, g1, g2)
    hs_p, hn_p, gr_p = run(x[:, perm].contiguous(), g1[:, perm].contiguous(), g2[perm].contiguous())
    assert hb.lstm_sync_status() == 0
    assert torch.equal(hs[:, perm], hs_p) and torch.equal(hn[perm], hn_p)
    for a, b, n in zip(gr, gr_p, names):
        close(b, a, rtol=2e-3, what="permutation invariance of d" + n)


def _disc_rows_f64(q, t, idx, rows):
    """Direct form of simple_fhvae.py:119-122 in float64 for a few queries: (CE terms, dq) of those rows."""
    qd, td = q[rows].double().cpu(), t.double().cpu()
    # |q - t|^2 through the expanded form IN FLOAT64 (1e-16 relative on ~1e2: exact for this purpose; the (rows, S, D)
    # difference tensor of the literal form would be 12 GB at S = 10^6)
    lg = -((qd * qd).sum(1)[:, None] - 2.0 * qd @ td.T + (td * td).sum(1)[None, :]) * float(hb_c())
    lse = torch.logsumexp(lg, 1)
    ar = torch.arange(len(rows))
    tgt = lg[ar, idx[rows].cpu()]
    p = torch.exp(lg - lse[:, None])
    p[ar, idx[rows].cpu()] -= 1.0
    dq = (-2.0 * hb_c()) * (qd * p.sum(1)[:, None] - p @ td)
    return lse - tgt, dq


def hb_c():
    import hip_binding

    return hip_binding.INV_TWO_VAR


# what ONE RANK of the 8-GPU configurations runs K5 on (dist_shard._ShardTable: every rank scans its rows for ALL global
# queries: 8 x 2048 = 16384), and configs[4]'s single-GPU table; (B, S, bf16 mode, workspace: None = the library's recommendation)
@pytest.mark.parametrize("B,S,lp", [(16384, 125000, False), (16384, 12500, True), (2048, 1000000, False)])
def test_disc_at_rank_view_shapes(hb, B, S, lp):
    """Translation invariance, shard associativity, one pass == two passes, and a float64 direct-form check of 48 sampled
    queries, at the shapes the first multi-GPU run will launch (K5 had never run with more than 2048 queries,
    nor the one-pass backward's partial buffers beyond S = 125 k)."""
    D = 32
    g = torch.Generator().manual_seed(B + S)
    q = torch.randn(B, D, generator=g).cuda()
    t = torch.randn(S, D, generator=g).cuda()
    idx = torch.randint(0, S, (B,), generator=g).cuda()
    rmax, rsum, tgt, ce = hb.raw_disc_fwd(q, t, idx, lp=lp)
    gs = torch.ones(1, device="cuda")
    dq1, dt1 = hb.raw_disc_bwd(q, t, idx, rmax, rsum, gs, 1.0 / B, lp=lp)               # one pass (whole problem or query groups)
    dq2, dt2 = hb.raw_disc_bwd(q, t, idx, rmax, rsum, gs, 1.0 / B, lp=lp, ws_bytes=0)   # two passes
    tol = 2e-3 if lp else 2e-5
    for a, b, what in ((dq1, dq2, "dq"), (dt1, dt2, "dtable")):
        assert torch.isfinite(a).all()
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert err <= tol * scale, (what, err, scale)
    resid = dq1.sum(0) + dt1.sum(0)   # logits depend on q - t only
    assert (resid.abs() <= (2e-2 if lp else 5e-4) * dq1.abs().sum(0) + 1e-7).all(), resid.abs().max().item()
    # three row shards combine to the unsharded statistics
    cuts = [0, S // 3, S // 3 + 1, S]
    parts = torch.empty(len(cuts) - 1, 3, B, device="cuda")
    for w, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        hb.raw_disc_fwd(q, t[a:b].contiguous(), idx, row0=a, want_ce=False, out3=parts[w], lp=lp)
    m, s, tg = hb.disc_merge_partials(parts)
    close(torch.log(s) + m, torch.log(rsum) + rmax, rtol=1e-5 if not lp else 1e-4, what="combined lse")
    close(hb.raw_disc_ce_mean(m, s, tg), ce, rtol=1e-5 if not lp else 1e-4, what="combined CE")
    # float64 direct form on sampled queries
    rows = torch.randint(0, B, (48,), generator=g)
    ce_rows, dq_rows = _disc_rows_f64(q, t, idx, rows)
    got = ((rmax - tgt) + torch.log(rsum))[rows.cuda()].double().cpu()
    assert (got - ce_rows).abs().max().item() <= (2e-3 if lp else 1e-4) * ce_rows.abs().max().item()
    err = (dq1[rows.cuda()].double().cpu() * B - dq_rows).abs().max().item()
    assert err <= (1e-2 if lp else 2e-4) * dq_rows.abs().max().item(), err


@pytest.mark.parametrize("lp", [False, True])
def test_disc_one_pass_in_query_groups(hb, lp):
    """The one-pass K5 backward with a workspace smaller than the whole problem needs takes the queries in groups of as many
    256-query tiles as fit (the partial buffers were unbounded); any size from one tile's worth up gives the
    gradients of the whole-problem call, less than one tile's worth falls back to two passes."""
    B, S, D = 1280, 9000, 32
    g = torch.Generator().manual_seed(7)
    q, t = torch.randn(B, D, generator=g).cuda(), torch.randn(S, D, generator=g).cuda()
    idx = torch.randint(0, S, (B,), generator=g).cuda()
    rmax, rsum, _, _ = hb.raw_disc_fwd(q, t, idx, lp=lp)
    gs = torch.full((1,), 0.7, device="cuda")
    lib = hb.load_library()
    full = int(lib.fhvae_disc_lse_bwd_ws_bytes(B, S, D))
    one_tile = (S * (D + 1) + (S // 64 + 64) * 256 * D) * 4 + 4096   # one tile's partials, generously
    ref = hb.raw_disc_bwd(q, t, idx, rmax, rsum, gs, 1.0 / B, lp=lp)
    for nbytes in (one_tile, 2 * one_tile, full // 2 // 16 * 16, 4096):
        assert nbytes < full
        got = hb.raw_disc_bwd(q, t, idx, rmax, rsum, gs, 1.0 / B, lp=lp, ws_bytes=nbytes)
        for a, b, what in zip(got, ref, ("dq", "dtable")):
            err = (a - b).abs().max().item()
            assert err <= (2e-3 if lp else 2e-5) * b.abs().max().item(), (nbytes, what, err)
    # dtable ACCUMULATES over the groups into whatever the caller hands in
    sink = torch.ones(S, D, device="cuda")
    hb.raw_disc_bwd(q, t, idx, rmax, rsum, gs, 1.0 / B, lp=lp, dt_sink=sink, ws_bytes=one_tile)
    assert ((sink - 1.0) - ref[1]).abs().max().item() <= (2e-3 if lp else 2e-5) * ref[1].abs().max().item() + 1e-6


def test_disc_sign_rides_in_the_kernels(hb):
    """log_qy = -CE (the intended objective) comes out of the K5 launches themselves: value and both gradients flip sign."""
    B, S, D = 512, 4600, 32
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
    t = torch.randn(S, D, generator=g).cuda().requires_grad_(True)
    idx = torch.randint(0, S, (B,), generator=g).cuda()
    ce = hb.disc_lse(q, t, idx)
    ce.backward()
    gq, gt = q.grad.clone(), t.grad.clone()
    q.grad = t.grad = None
    neg = hb.disc_lse(q, t, idx, sign=-1.0)
    neg.backward()
    assert torch.equal(neg.detach(), -ce.detach())
    close(q.grad, -gq, rtol=1e-6, what="dq of -CE")
    close(t.grad, -gt, rtol=1e-5, what="dtable of -CE")
