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
