"""GPU parity tests of each C-ABI op (through hip_binding -> libfhvae_hip.so) against the CPU oracle
(oracle/ref_cpu.py) on the same seeded inputs.  Tolerance: 1e-4 relative fp32 (BASELINE.json north_star),
applied as rtol=1e-4 with an absolute floor of 1e-4 x the tensor's max magnitude."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R


@pytest.fixture(scope="module")
def hb():
    import build_ext

    build_ext.build(verbose=False)
    import hip_binding

    hip_binding.load_library()
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return hip_binding


def close(got, want, rtol=1e-4, what=""):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(want.abs().max().item(), 1e-30)
    torch.testing.assert_close(got, want, rtol=rtol, atol=rtol * scale, msg=lambda m: "%s: %s" % (what, m))


def dev(t):
    return t.cuda()


@pytest.mark.parametrize("M,K,N,relu", [(8, 48, 16, True), (64, 1600, 128, True), (256, 1616, 128, False), (5, 7, 3, True),
                                        (300, 130, 70, False), (64, 32, 1600, False)])
def test_linear_fwd_bwd(hb, M, K, N, relu):
    torch.manual_seed(M + K)
    x = torch.randn(M, K, requires_grad=True)
    w = (torch.randn(N, K) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, requires_grad=True)
    y = torch.nn.functional.linear(x, w, b)
    y = torch.relu(y) if relu else y
    g = torch.randn(M, N)
    y.backward(g)
    xd, wd, bd = (dev(t.detach()).requires_grad_(True) for t in (x, w, b))
    yd = hb.linear(xd, wd, bd, relu)
    yd.backward(dev(g))
    close(yd, y, what="y")
    close(xd.grad, x.grad, what="dx")
    close(wd.grad, w.grad, what="dw")
    close(bd.grad, b.grad, what="db")


def test_gauss_head(hb):
    torch.manual_seed(1)
    M, K, D = 37, 96, 16
    h = torch.randn(M, K, requires_grad=True)
    wm, wl = (torch.randn(D, K) * 0.1).requires_grad_(True), (torch.randn(D, K) * 0.1).requires_grad_(True)
    bm, bl = torch.randn(D, requires_grad=True), torch.randn(D, requires_grad=True)
    eps = torch.randn(M, D)
    mu, lv = torch.nn.functional.linear(h, wm, bm), torch.nn.functional.linear(h, wl, bl)
    s = R.gauss_sample(mu, lv, eps)
    gm, gl, gs = torch.randn(M, D), torch.randn(M, D), torch.randn(M, D)
    (mu * gm + lv * gl + s * gs).sum().backward()
    args = [dev(t.detach()).requires_grad_(True) for t in (h, wm, bm, wl, bl)]
    mud, lvd, sd = hb.gauss_head(*args, dev(eps))
    (mud * dev(gm) + lvd * dev(gl) + sd * dev(gs)).sum().backward()
    for got, want, n in ((mud, mu, "mu"), (lvd, lv, "lv"), (sd, s, "sample")):
        close(got, want, what=n)
    for a, r, n in zip(args, (h, wm, bm, wl, bl), "h wm bm wl bl".split()):
        close(a.grad, r.grad, what="d" + n)
    # no-sample form (decoder head)
    mud2, lvd2, none = hb.gauss_head(*[a.detach() for a in args], None)
    assert none is None
    close(mud2, mu, what="mu(nosample)")


@pytest.mark.parametrize("M,K,D,sample", [(37, 96, 16, True), (2048, 512, 32, True), (4100, 256, 80, False), (300, 64, 8, False),
                                          (1000, 128, 40, True)])
def test_gauss_head_bf16_operands(hb, M, K, D, sample):
    """The bf16-operand head (hip_binding._GaussHeadLp: stacked weights, one projection each way, csrc/proj.hip + wgrad.hip where
    the shape allows, the generic engine otherwise) against torch autograd on the SAME bf16-rounded operands in f64: outputs
    1e-4 of their scale (f32 accumulation), gradients 2e-2 (the upstream gradient is itself rounded to bf16)."""
    torch.manual_seed(M + K)
    rb = lambda t: t.bfloat16().double()
    h = torch.randn(M, K)
    wm, wl = torch.randn(D, K) * 0.1, torch.randn(D, K) * 0.1
    bm, bl = torch.randn(D), torch.randn(D)
    eps = torch.randn(M, D) if sample else None
    gm, gl, gs = torch.randn(M, D), torch.randn(M, D), torch.randn(M, D)
    ref = [t.requires_grad_(True) for t in (rb(h), rb(wm), bm.double(), rb(wl), bl.double())]
    mu, lv = torch.nn.functional.linear(ref[0], ref[1], ref[2]), torch.nn.functional.linear(ref[0], ref[3], ref[4])
    tot = (mu * gm.double()).sum() + (lv * gl.double()).sum()
    if sample:
        smp = mu + eps.double() * torch.exp(0.5 * lv)
        tot = tot + (smp * gs.double()).sum()
    tot.backward()
    args = [dev(t).requires_grad_(True) for t in (h, wm, bm, wl, bl)]
    mud, lvd, sd = hb.gauss_head(*args, dev(eps) if sample else None, h_lp=dev(h).bfloat16())
    totd = (mud * dev(gm)).sum() + (lvd * dev(gl)).sum()
    if sample:
        totd = totd + (sd * dev(gs)).sum()
    totd.backward()
    outs = [(mud, mu, "mu"), (lvd, lv, "lv")] + ([(sd, smp, "sample")] if sample else [])
    for got, want, n in outs:
        assert _max_rel(got, want) < 1e-4, (n, _max_rel(got, want))
    for a, r, n in zip(args, ref, "h wm bm wl bl".split()):
        assert _max_rel(a.grad, r.grad) < 2e-2, ("d" + n, _max_rel(a.grad, r.grad))


def test_head_takes_the_lower_bound_kernels_bf16_gradient(hb):
    """Per-frame head + lower bound, time-major (T*B, F) rows: fhvae_elbo_bwd leaves [d_x_mu | d_x_lv] in bf16 with the column
    sums, and the head's backward takes them (no second pass over the f32 gradients).  Same parameter gradients and dh as with
    the hand-over switched off (the head then rounds the f32 gradients itself: identical bf16 operand; the bias gradients are
    sums of the f32 values in one case and of the rounded ones in the other: 2e-3)."""
    torch.manual_seed(3)
    T, B, F, K, D2 = 20, 96, 80, 256, 16
    x = dev(torch.randn(T, B, F))
    h = dev(torch.randn(T * B, K))
    w = [dev(t) for t in (torch.randn(F, K) * 0.05, torch.randn(F) * 0.1, torch.randn(F, K) * 0.05, torch.randn(F) * 0.1)]
    z = [dev(torch.randn(B, D2)) for _ in range(5)]
    ns = dev(torch.randint(20, 100, (B,)))
    layout = (B, T, F, (F, B * F), (F, B * F))

    def run(side):
        hb.PAIR_SIDE["enabled"] = side
        used0 = hb.PAIR_SIDE["used"]
        try:
            hh = h.clone().requires_grad_(True)
            ws = [t.clone().requires_grad_(True) for t in w]
            x_mu, x_lv, _ = hb.gauss_head(hh, *ws, None, h_lp=hh.detach().bfloat16())
            lb = hb.elbo(x, x_mu, x_lv, *z, ns, layout, False)[0]
            (lb * dev(torch.linspace(0.5, 1.5, B))).sum().backward()
            torch.cuda.synchronize()
            assert hb.PAIR_SIDE["used"] - used0 == (1 if side else 0)
            return [hh.grad] + [t.grad for t in ws]
        finally:
            hb.PAIR_SIDE["enabled"] = True

    a, b = run(True), run(False)
    for u, v, n in zip(a, b, "dh dwm dbm dwl dbl".split()):
        assert _max_rel(u, v) < 2e-3, (n, _max_rel(u, v))


def _max_rel(got, want):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,T,I,Ic,H,L,dt", [(5, 4, 6, 0, 8, 2, "f32"), (5, 4, 6, 4, 8, 2, "f32"), (7, 3, 0, 8, 16, 1, "f32"),
                                             (70, 20, 80, 0, 64, 2, "f32"), (33, 20, 80, 32, 48, 2, "f32"),
                                             (64, 20, 0, 64, 256, 2, "f32"), (300, 20, 80, 32, 256, 2, "f32"),
                                             (5, 4, 8, 8, 8, 2, "bf16"), (70, 20, 80, 0, 64, 2, "bf16"),
                                             (33, 20, 80, 32, 48, 2, "bf16"), (64, 20, 0, 64, 256, 2, "bf16"),
                                             (300, 20, 80, 32, 256, 2, "bf16")])
def test_lstm_seq_vs_torch_lstm(hb, B, T, I, Ic, H, L, dt):
    """f32 mode: 1e-4 (exact-f32 MFMA).  bf16 mode: operands rounded to 8 mantissa bits, f32 accumulation and
    cell state; stated tolerance 3e-2 of the tensor's max magnitude forward, 6e-2 for gradients."""
    torch.manual_seed(B * 31 + H)
    dtype = hb.BF16 if dt == "bf16" else hb.F32
    tol_f, tol_g = (3e-2, 6e-2) if dt == "bf16" else (1e-4, 1e-4)
    lstm = torch.nn.LSTM(I + Ic, H, L, batch_first=True)
    x = torch.randn(B, T, I) if I else None
    xc = torch.randn(B, Ic, requires_grad=True) if Ic else None
    parts = ([x] if I else []) + ([xc[:, None, :].expand(B, T, Ic)] if Ic else [])
    out, (hn, _) = lstm(torch.cat(parts, -1))
    hn_cat = torch.cat([hn[l] for l in range(L)], -1)
    g_out, g_hn = torch.randn(B, T, H), torch.randn(B, L * H)
    ((out * g_out).sum() + (hn_cat * g_hn).sum()).backward()

    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    params = [dev(getattr(lstm, n).detach()).requires_grad_(True) for n in names]
    x_tm = dev(x.transpose(0, 1).contiguous()) if I else None
    xcd = dev(xc.detach()).requires_grad_(True) if Ic else None
    hs_top, hnd = hb.lstm_seq(x_tm, xcd, T, params, dtype)
    ((hs_top * dev(g_out.transpose(0, 1).contiguous())).sum() + (hnd * dev(g_hn)).sum()).backward()
    close(hs_top.transpose(0, 1), out, rtol=tol_f, what="hs_top")
    close(hnd, hn_cat, rtol=tol_f, what="hn")
    worst = 0.0
    for p, n in zip(params, names):
        worst = max(worst, _max_rel(p.grad, getattr(lstm, n).grad))
        close(p.grad, getattr(lstm, n).grad, rtol=tol_g, what="d" + n)
    if Ic:
        close(xcd.grad, xc.grad, rtol=tol_g, what="dxc")
    print("lstm %s B=%d H=%d: fwd max-rel %.2e, worst grad max-rel %.2e" % (dt, B, H, _max_rel(hnd, hn_cat), worst))


@pytest.mark.parametrize("K,M,N", [(40960, 1024, 256), (1280, 512, 80), (5000, 192, 112), (777, 1024, 256), (8192, 2048, 512),
                                   (64, 256, 256), (4096 + 8, 320, 136)])
def test_wgrad_bf16_vs_cpu_matmul(hb, K, M, N):
    """fhvae_wgrad_bf16: C += A^T B over K rows (csrc/wgrad.hip: 256x256 / 256x128 tiles, LDS-DMA double buffer, hardware
    transposed fragment reads, split-K atomics), K tails / ragged M, N / odd slice counts included, against the f32 CPU matmul
    of the same bf16 operands.  f32 accumulation in a different order: 2e-5 of the output's scale sqrt(K)."""
    torch.manual_seed(K + M)
    # padded leading dimensions on some cases: the operands are column ranges of wider matrices in the model
    lda, ldb = M + (8 if K % 2 else 0), N + (16 if M % 3 == 0 else 0)
    a_full, b_full = torch.randn(K, lda).bfloat16(), torch.randn(K, ldb).bfloat16()
    a, b = a_full[:, :M], b_full[:, :N]
    c0 = torch.randn(M, N)
    want = c0.double() + a.double().t() @ b.double()
    cd = dev(c0)
    hb.wgrad_bf16_(cd, dev(a_full)[:, :M], dev(b_full)[:, :N])
    err = (cd.cpu().double() - want).abs().max().item()
    assert err <= 2e-5 * K ** 0.5 + 1e-6, (err, K)
    # accumulates: a second call adds the product again
    hb.wgrad_bf16_(cd, dev(a_full)[:, :M], dev(b_full)[:, :N])
    want2 = want + a.double().t() @ b.double()
    assert (cd.cpu().double() - want2).abs().max().item() <= 4e-5 * K ** 0.5 + 1e-6


@pytest.mark.parametrize("K,M,N", [(20480, 1024, 256), (1280, 512, 80), (5000, 192, 112), (777, 1024, 256), (4096, 2048, 512),
                                   (32, 256, 256), (4096 + 4, 320, 136)])
def test_wgrad_f32_vs_cpu_matmul(hb, K, M, N):
    """fhvae_wgrad_f32: the exact-f32 form of the long-K weight-gradient kernel (csrc/wgrad_f32.hip: 256x256 / 256x128 tiles, LDS-DMA
    double buffer, one ds_read_b32 per operand scalar, split-K atomics), K tails / ragged M, N / odd slice counts included, against the
    f64 CPU matmul of the same f32 operands.  Exact-f32 products; what remains is the f32 accumulation (a random walk of K
    roundings at the partial sum's magnitude): 1e-5 of the output's scale sqrt(K), the bound of the bf16 kernel's test."""
    torch.manual_seed(K + M)
    lda, ldb = M + (4 if K % 2 else 0), N + (8 if M % 3 == 0 else 0)
    a_full, b_full = torch.randn(K, lda), torch.randn(K, ldb)
    a, b = a_full[:, :M], b_full[:, :N]
    c0 = torch.randn(M, N)
    want = c0.double() + a.double().t() @ b.double()
    cd = dev(c0)
    hb.wgrad_f32_(cd, dev(a_full)[:, :M], dev(b_full)[:, :N])
    err = (cd.cpu().double() - want).abs().max().item()
    assert err <= 1e-5 * K ** 0.5 + 1e-6, (err, K)
    hb.wgrad_f32_(cd, dev(a_full)[:, :M], dev(b_full)[:, :N])  # accumulates
    want2 = want + a.double().t() @ b.double()
    assert (cd.cpu().double() - want2).abs().max().item() <= 2e-5 * K ** 0.5 + 1e-6


@pytest.mark.parametrize("M,N,K", [(40960, 256, 1024), (40960, 160, 256), (2048, 64, 512), (1000, 256, 64), (300, 96, 128), (17, 4, 64),
                                   (5000, 512, 320)])
def test_proj_bf16_vs_cpu_matmul(hb, M, N, K):
    """fhvae_proj_bf16: C = A B^T (+ bias) with K-contiguous bf16 operands (csrc/proj.hip: one row tile per CU, LDS-DMA double
    buffer, swizzled ds_read_b128 fragments), ragged M / N, two column tiles, padded leading dimensions, against the f64 CPU
    matmul of the same bf16 operands.  f32 accumulation in a different order: 2e-5 of the output's scale sqrt(K)."""
    torch.manual_seed(M + N)
    lda, ldb = K + (8 if M % 2 else 0), K + (16 if N % 3 == 0 else 0)
    a_full, b_full = torch.randn(M, lda).bfloat16(), torch.randn(N, ldb).bfloat16()
    a, b = a_full[:, :K], b_full[:, :K]
    bias = torch.randn(N) if M % 3 else None
    want = a.double() @ b.double().t() + (bias.double() if bias is not None else 0.0)
    out = torch.full((M, N + 4), 7.0, device="cuda")  # a column range of a wider matrix: the pad must stay untouched
    got = hb.proj_bf16(dev(a_full)[:, :K], dev(b_full)[:, :K], dev(bias) if bias is not None else None, out=out[:, :N])
    err = (got.cpu().double() - want).abs().max().item()
    assert err <= 2e-5 * K ** 0.5 + 1e-6, (err, K)
    assert (out[:, N:] == 7.0).all()


def test_proj_bf16_rejects_unaligned(hb):
    a, b = torch.zeros(64, 72, device="cuda").bfloat16(), torch.zeros(8, 72, device="cuda").bfloat16()
    with pytest.raises(RuntimeError):
        hb.proj_bf16(a, b)  # K = 72 is not a multiple of 64


def test_deferred_param_grads_match_immediate(hb):
    """With gradient sinks (FusedAdam) the nets' parameter gradients are queued and flushed as one grouped call by the
    optimizer: same gradients as the immediate path (FHVAE_NO_DEFER semantics via set_defer_param_grads(False))."""
    from fhvae import FHVAE
    from hip_optim import FusedAdam
    from train_model import loss_function

    T, F, H, D, B, S = 20, 80, 128, 16, 128, 50
    g = torch.Generator().manual_seed(9)
    x, idx, ns = torch.randn(B, T, F, generator=g).cuda(), torch.randint(0, S, (B,), generator=g), torch.randint(20, 200, (B,), generator=g)
    eps = (torch.randn(B, D, generator=g).cuda(), torch.randn(B, D, generator=g).cuda())
    grads = []
    for defer in (True, False):
        hb.set_defer_param_grads(defer)
        try:
            torch.manual_seed(4)
            m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T, num_seqs=S, reference_compat=False, compute_dtype="bf16").cuda()
            opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
            opt.zero_grad()
            out = m(x, idx, S, ns, eps=eps)
            loss_function(out[0], out[1], 10.0).backward()
            assert bool(hb._DEFER["pending"]) == defer
            grads.append(opt.flat_grad().clone())
            assert not hb._DEFER["pending"]
        finally:
            hb.set_defer_param_grads(True)
    scale = grads[1].abs().max().item()
    assert (grads[0] - grads[1]).abs().max().item() <= 1e-4 * scale, ((grads[0] - grads[1]).abs().max().item(), scale)


def test_deferred_branch_fires_the_recurrence_hook_and_accumulates_twice(hb):
    """(a) the deferred weight-gradient branch fires hip_binding.LSTM_BWD_REC_HOOK once per net (the distributed
    runner's overlap hangs on it); (b) two backward passes queued before ONE flush put the same weight matrix twice into one
    grouped launch (wgrad.hip: shared_c -> atomics, no plain read-modify-write race): the flushed gradient is twice a single pass's."""
    from fhvae import FHVAE
    from hip_optim import FusedAdam
    from train_model import loss_function

    T, F, H, D, B, S = 20, 80, 128, 16, 128, 50
    g = torch.Generator().manual_seed(11)
    x, idx, ns = torch.randn(B, T, F, generator=g).cuda(), torch.randint(0, S, (B,), generator=g), torch.randint(20, 200, (B,), generator=g)
    eps = (torch.randn(B, D, generator=g).cuda(), torch.randn(B, D, generator=g).cuda())
    torch.manual_seed(5)
    m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T, num_seqs=S, reference_compat=False, compute_dtype="bf16").cuda()
    opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
    fired = []
    hb.LSTM_BWD_REC_HOOK["fn"] = lambda sinks: fired.append(len(sinks))
    try:
        opt.zero_grad()
        out = m(x, idx, S, ns, eps=eps)
        loss_function(out[0], out[1], 10.0).backward()
        assert len(fired) == 3 and len(hb._DEFER["pending"]) == 3  # three nets, each queued behind its recurrence
        once = opt.flat_grad().clone()
        opt.zero_grad()
        for _ in range(2):
            out = m(x, idx, S, ns, eps=eps)
            loss_function(out[0], out[1], 10.0).backward()
        assert len(hb._DEFER["pending"]) == 6
        twice = opt.flat_grad().clone()
    finally:
        hb.LSTM_BWD_REC_HOOK["fn"] = None
    scale = once.abs().max().item()
    assert (twice - 2 * once).abs().max().item() <= 2e-4 * scale, ((twice - 2 * once).abs().max().item(), scale)


def test_backward_refuses_a_forward_of_another_schedule(hb):
    """the schedule (and with it the layout of the saved gates) is re-derived from the environment per call; a switch
    flipped between forward and backward must raise, not return gradients computed from misread gates."""
    B, T, I, H, L = 1024, 3, 80, 256, 2
    torch.manual_seed(2)
    lstm = torch.nn.LSTM(I, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    ps = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
    x = torch.randn(T, B, I).cuda()
    hs_top, hn = hb.lstm_seq(x, None, T, ps, hb.BF16)
    os.environ["FHVAE_NO_FWD_WR"] = "1"  # the backward would now expect row-major gates
    try:
        with pytest.raises(RuntimeError):
            (hs_top.sum() + hn.sum()).backward()
    finally:
        os.environ.pop("FHVAE_NO_FWD_WR", None)
        hb.flush_param_grads()


@pytest.mark.parametrize("B,T,F", [(6, 5, 12), (7, 3, 10), (300, 20, 80)])
def test_to_time_major(hb, B, T, F):
    """both forms of the copy: four features per thread (F % 4 == 0) and the scalar one; with the bf16 copy riding along"""
    x = torch.randn(B, T, F)
    want = x.transpose(0, 1).contiguous()
    assert torch.equal(hb.to_time_major(dev(x)).cpu(), want)
    both = hb.to_time_major(dev(x), with_bf16=True)
    assert torch.equal(both.cpu(), want) and torch.equal(both._fh_lp.cpu(), want.bfloat16())


@pytest.mark.parametrize("detach", [True, False])
@pytest.mark.parametrize("B,T,F,D,tm", [(8, 4, 8, 16, False), (64, 20, 80, 32, True), (5, 3, 7, 4, False), (260, 20, 80, 16, True)])
def test_elbo_fwd_bwd(hb, B, T, F, D, tm, detach):
    torch.manual_seed(B + D)
    x = torch.randn(B, T, F)
    xm = torch.randn(B, T, F, requires_grad=True)
    xl = (torch.randn(B, T, F) * 0.5).requires_grad_(True)
    z = [torch.randn(B, D, requires_grad=True) for _ in range(5)]  # z1_mu z1_lv z2_mu z2_lv mu2
    ns = torch.randint(3, 200, (B,))
    outs = R.elbo_terms(x, xm, xl, z[0], z[1], z[2], z[3], z[4], ns, reference_detach=detach)
    gw = [torch.randn(B) for _ in range(5)]
    sum((o * g).sum() for o, g in zip(outs, gw) if o.requires_grad).backward()

    def lay(t):  # batch-major (B,T,F) or time-major (T,B,F) copy on the device
        return dev(t.detach().transpose(0, 1).contiguous() if tm else t.detach())

    xd, xmd, xld = lay(x), lay(xm).requires_grad_(True), lay(xl).requires_grad_(True)
    zd = [dev(t.detach()).requires_grad_(True) for t in z]
    strides = (F, B * F) if tm else (T * F, F)
    got = hb.elbo(xd, xmd, xld, *zd, dev(ns), (B, T, F, strides, strides), detach)
    order = [0, 1, 2, 3, 4]  # both return (lower_bound, log_px_z, neg_kld_z1, neg_kld_z2, log_pmu2)
    for k in order:
        close(got[k], outs[k], what="out%d" % k)
    assert got[1].requires_grad == (not detach) and got[4].requires_grad == (not detach)
    sum((o * dev(g)).sum() for o, g in zip(got, gw) if o.requires_grad).backward()
    for a, r, n in zip(zd, z, "z1_mu z1_lv z2_mu z2_lv mu2".split()):
        close(a.grad, r.grad, what="d" + n)
    if detach:
        assert xmd.grad is None and xld.grad is None and xm.grad is None
    else:
        unlay = (lambda t: t.transpose(0, 1)) if tm else (lambda t: t)
        close(unlay(xmd.grad), xm.grad, what="dx_mu")
        close(unlay(xld.grad), xl.grad, what="dx_lv")


def test_elbo_scalar_nsegs(hb):
    torch.manual_seed(0)
    B, T, F, D = 6, 4, 8, 16
    x, xm, xl = torch.randn(B, T, F), torch.randn(B, T, F), torch.randn(B, T, F) * 0.3
    z = [torch.randn(B, D) for _ in range(5)]
    want = R.elbo_terms(x, xm, xl, *z, 17, reference_detach=True)
    got = hb.elbo(dev(x), dev(xm), dev(xl), *[dev(t) for t in z], 17, (B, T, F, (T * F, F), (T * F, F)), True)
    for k in range(5):
        close(got[k], want[k], what="out%d" % k)


def test_mu2_gather_with_collisions(hb):
    torch.manual_seed(0)
    S, D, B = 12, 16, 40
    table = torch.randn(S, D, requires_grad=True)
    idx = torch.randint(0, S, (B,))
    idx[:8] = 3  # heavy collision on one row
    g = torch.randn(B, D)
    R.mu2_gather(table, idx).backward(g)
    td = dev(table.detach()).requires_grad_(True)
    out = hb.mu2_gather(td, dev(idx))
    out.backward(dev(g))
    close(out, table.detach()[idx], rtol=0, what="gather")
    close(td.grad, table.grad, what="dtable")


@pytest.mark.parametrize("name", ["disc_8x12.npz", "disc_256x4600.npz"])
def test_disc_lse_golden(hb, golden_dir, name):
    g = dict(np.load(os.path.join(golden_dir, name)))
    q = dev(torch.from_numpy(g["q"])).requires_grad_(True)
    t = dev(torch.from_numpy(g["table"])).requires_grad_(True)
    idx = dev(torch.from_numpy(g["idx"]))
    v = hb.disc_lse(q, t, idx)
    close(v, torch.from_numpy(g["log_qy"]), what="log_qy")
    v.backward()
    close(q.grad, torch.from_numpy(g["dq"]), what="dq")
    close(t.grad, torch.from_numpy(g["dtable"]), what="dtable")


@pytest.mark.parametrize("B,S,D,spread", [(3, 5, 4, 1.0), (300, 1000, 32, 1.0), (257, 777, 16, 6.0), (64, 9, 8, 3.0),
                                          (2048, 4600, 32, 1.0), (513, 3001, 32, 4.0), (70, 1000, 32, 0.3)])
def test_disc_lse_vs_oracle(hb, B, S, D, spread):
    torch.manual_seed(B + S)
    q = (torch.randn(B, D) * spread).requires_grad_(True)
    t = torch.randn(S, D, requires_grad=True)
    idx = torch.randint(0, S, (B,))
    want = R.disc_loss(q, t, idx)
    (want * 1.7).backward()
    qd, td = dev(q.detach()).requires_grad_(True), dev(t.detach()).requires_grad_(True)
    got = hb.disc_lse(qd, td, dev(idx))
    (got * 1.7).backward()
    close(got, want, what="log_qy")
    close(qd.grad, q.grad, what="dq")
    close(td.grad, t.grad, what="dtable")


def test_disc_lse_sharded_partials_combine(hb):
    """Row-sharded table: per-shard (max, sumexp, target) partials combine to the unsharded value
    (the multi-GPU exchange of SURVEY 8e, exercised on one GPU)."""
    torch.manual_seed(5)
    B, S, D = 130, 1000, 32
    q, t = torch.randn(B, D) * 2, torch.randn(S, D)
    idx = torch.randint(0, S, (B,))
    want = R.disc_loss(q, t, idx)
    qd, idxd = dev(q), dev(idx)
    cuts = [0, 250, 251, 700, 1000]
    ms, ss, tg = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rmax, rsum, tgt, _ = hb.raw_disc_fwd(qd, dev(t[a:b].contiguous()), idxd, row0=a, want_ce=False)
        ms.append(rmax.cpu()), ss.append(rsum.cpu()), tg.append(tgt.cpu())
    m = torch.stack(ms).max(0).values
    s = sum(si * torch.exp(mi - m) for si, mi in zip(ss, ms))
    ce = ((m - sum(tg)) + torch.log(s)).mean()
    close(ce, want, what="sharded CE")


def test_adam_matches_torch(hb):
    torch.manual_seed(0)
    n = 1000
    p = torch.randn(n)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.95, 0.999))
    pd, m, v = dev(p.clone()), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    step = torch.zeros((), dtype=torch.int32).cuda()
    for k in range(5):
        g = torch.randn(n)
        ref.grad = g.clone()
        opt.step()
        step += 1
        hb.adam_step_(pd, dev(g), m, v, step, 1e-3, 0.95, 0.999, 1e-8)
    close(pd, ref, rtol=1e-5, what="adam p")


@pytest.mark.parametrize("n", [1000, 103427, 3 * 8192 * 1024 + 5])
def test_adam_counts_its_step_and_clears_the_gradient(hb, n):
    """ABI 11: FHVAE_ADAM_ADVANCE (the launch uses step[0] + 1 and stores it; the last workgroup, so nobody reads the new value)
    and FHVAE_ADAM_ZERO_GRAD (g cleared behind its use) give the same parameters as the increment-then-launch form -- also when
    the grid is at its cap and grid-strides (the large case)."""
    torch.manual_seed(n % 97)
    p0 = torch.randn(n, device="cuda")
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device="cuda") for _ in range(4))
    step_a = torch.zeros((), dtype=torch.int32, device="cuda")
    step_b = torch.zeros(hb.ADAM_STEP_WORDS, dtype=torch.int32, device="cuda")
    for k in range(3):
        g = torch.randn(n, device="cuda")
        ga, gb = g.clone(), g.clone()
        step_a += 1
        hb.adam_step_(pa, ga, ma, va, step_a, 1e-3, 0.95, 0.999, 1e-8)
        hb.adam_step_(pb, gb, mb, vb, step_b, 1e-3, 0.95, 0.999, 1e-8, flags=hb.ADAM_ZERO_GRAD | hb.ADAM_ADVANCE)
        assert torch.equal(ga, g) and float(gb.abs().max()) == 0.0
        assert step_b.tolist() == [k + 1] + [0] * (hb.ADAM_STEP_WORDS - 1)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    with pytest.raises(RuntimeError):
        hb.adam_step_(pb, gb, mb, vb, step_a, 1e-3, 0.95, 0.999, 1e-8, flags=hb.ADAM_ADVANCE)


def test_fused_adam_arena_matches_torch_adam(hb):
    from hip_optim import FusedAdam

    torch.manual_seed(1)
    shapes = [(7, 5), (13,), (64, 3)]
    ps = [torch.randn(s) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=1e-3, betas=(0.95, 0.999))
    o_mine = FusedAdam(mine, lr=1e-3, betas=(0.95, 0.999))
    for k in range(4):
        o_ref.zero_grad()
        o_mine.zero_grad()
        gs = [torch.randn(s) for s in shapes]
        sum((p * g).sum() for p, g in zip(ref, gs)).backward()
        sum((p * g.cuda()).sum() for p, g in zip(mine, gs)).backward()  # autograd accumulates into the arena views
        o_ref.step()
        o_mine.step()
    for a, b in zip(mine, ref):
        close(a, b, rtol=1e-5, what="fused adam")
    assert all(p.data_ptr() == v.data_ptr() for p, v in zip(mine, o_mine.p_arena.views))
    # the Adam launch left the gradient arena zero and counted its step; zero_grad() after step() is then free, and a
    # zero_grad() that follows a backward without a step still clears
    assert int(o_mine.step_dev.item()) == 4 and float(o_mine.g_arena.flat.abs().max()) == 0.0 and o_mine._zeroed_by_step
    o_mine.zero_grad()
    assert not o_mine._zeroed_by_step
    sum((p * 2.0).sum() for p in mine).backward()
    assert float(o_mine.g_arena.flat.abs().max()) == 2.0
    o_mine.zero_grad()
    assert float(o_mine.g_arena.flat.abs().max()) == 0.0


def test_fused_loss_matches_expression(hb):
    """train_model.loss_function on the model's GPU outputs = fhvae_loss_fwd/bwd: same value and gradients as the
    reference expression -mean(lower_bound + alpha*log_qy) (train_model.py:243-251)."""
    import train_model

    torch.manual_seed(5)
    lb = torch.randn(777, device="cuda").mul_(50).requires_grad_(True)
    qy = torch.randn((), device="cuda").requires_grad_(True)
    loss = train_model.loss_function(lb, qy, 10.0)
    (loss * 3.0).backward()
    lb2, qy2 = lb.detach().clone().requires_grad_(True), qy.detach().clone().requires_grad_(True)
    want = -1 * torch.mean(lb2 + 10.0 * qy2)
    (want * 3.0).backward()
    close(loss, want, rtol=1e-5, what="loss")
    close(lb.grad, lb2.grad, rtol=1e-6, what="d_lb")
    close(qy.grad, qy2.grad, rtol=1e-6, what="d_qy")


@pytest.mark.gpu
def test_lstm_workspace_size_queries(hb):
    """fhvae_lstm_pre_elems / fhvae_lstm_ws_below_elems / fhvae_lstm_lp_bytes per schedule: the (T,B,4H) `pre` buffer only where a
    schedule reads it, `ws_below` only for the layer-by-layer persistent backward at H != 256, a workspace in f32 mode too."""
    import ctypes as C

    lib = hb.load_library()

    def desc(dtype, L, B, T, I, Ic, H):
        d = hb.LstmDesc()
        params = []
        for l in range(L):
            kin = I + Ic if l == 0 else H
            params += [torch.zeros(4 * H, kin, device="cuda"), torch.zeros(4 * H, H, device="cuda"), torch.zeros(4 * H, device="cuda"),
                       torch.zeros(4 * H, device="cuda")]
        hb._fill_lstm_desc(d, dtype, (L, B, T, I, Ic, H), None, None, params)
        n = int(lib.fhvae_lstm_lp_bytes(C.byref(d)))
        lp = torch.empty(max(n, 16), device="cuda", dtype=torch.uint8)
        d.lp = lp.data_ptr()
        return d, n, (params, lp)

    T = 20
    d, n, keep = desc(hb.BF16, 2, 2048, T, 80, 32, 256)  # persistent rows form, x folded into the kernel
    assert n > 0 and lib.fhvae_lstm_pre_elems(C.byref(d)) == 2048 * 4 * 256 and lib.fhvae_lstm_ws_below_elems(C.byref(d)) == T * 2048 * 256
    d, n, keep = desc(hb.BF16, 2, 2048, T, 80, 0, 128)  # persistent, H = 128: the from-above term goes through ws_below
    assert lib.fhvae_lstm_ws_below_elems(C.byref(d)) == T * 2048 * 128
    d, n, keep = desc(hb.BF16, 2, 2048, T, 80, 32, 512)  # large-tile cells: they multiply layer 0's input themselves
    assert lib.fhvae_lstm_pre_elems(C.byref(d)) == 1
    d, n, keep = desc(hb.BF16, 2, 64, T, 80, 32, 512)  # generic per-step cells
    assert lib.fhvae_lstm_pre_elems(C.byref(d)) == T * 64 * 4 * 512 and lib.fhvae_lstm_ws_below_elems(C.byref(d)) == 0
    d, n, keep = desc(hb.F32, 2, 64, T, 80, 32, 256)  # f32: workspace = sync block + transposed weights
    assert n == 16384 + 4 * (3 * 4 * 256 * 256) and lib.fhvae_lstm_pre_elems(C.byref(d)) == T * 64 * 4 * 256
    d, n, keep = desc(hb.F32, 2, 64, T, 0, 64, 256)  # time-constant input only: one (B,4H) slab
    assert lib.fhvae_lstm_pre_elems(C.byref(d)) == 64 * 4 * 256


@pytest.mark.parametrize("M,D", [(2048, 32), (100, 32), (40960, 80), (7, 8)])
def test_reparam_bwd_pair_bias_sums_ride_along(hb, M, D):
    """ABI 10: fhvae_gauss_reparam_bwd_pair adds the two bias gradients (column sums of the bf16 operand it writes) in the same
    launch where a workgroup covers whole rows (ldg/8 divides 256), and through the column-sum kernel otherwise."""
    lib = hb.load_library()
    g = torch.Generator().manual_seed(M + D)
    d_mu, d_lv, eps = (torch.randn(M, D, generator=g).cuda() for _ in range(3))
    # ABI 11: d_sample with a row stride -- a column slice of the next net's input gradient (cat's backward) as it is
    d_s = torch.randn(M, 2 * D + 8, generator=g).cuda()[:, D:2 * D]
    assert not d_s.is_contiguous()
    lv = (torch.randn(M, D, generator=g) * 0.3).cuda()
    ldg = (2 * D + 63) // 64 * 64
    g_lp = torch.empty(M, ldg, device="cuda", dtype=torch.bfloat16)
    db_mu, db_lv = torch.full((D,), 0.5, device="cuda"), torch.full((D,), -0.25, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.fhvae_gauss_reparam_bwd_pair(d_mu.data_ptr(), d_lv.data_ptr(), d_s.data_ptr(), d_s.stride(0), eps.data_ptr(), lv.data_ptr(), D,
                                            g_lp.data_ptr(), ldg, db_mu.data_ptr(), db_lv.data_ptr(), M, D, st) == 0
    want_mu = d_mu + d_s
    want_lv = d_lv + d_s * eps * 0.5 * torch.exp(0.5 * lv)
    close(g_lp[:, :D].float(), want_mu, rtol=1e-2, what="g_mu (bf16)")
    close(g_lp[:, D:2 * D].float(), want_lv, rtol=1e-2, what="g_lv (bf16)")
    assert float(g_lp[:, 2 * D:].float().abs().sum()) == 0.0
    close(db_mu - 0.5, g_lp[:, :D].double().sum(0).float(), rtol=1e-5, what="db_mu += column sums")
    close(db_lv + 0.25, g_lp[:, D:2 * D].double().sum(0).float(), rtol=1e-5, what="db_lv += column sums")
