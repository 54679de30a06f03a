"""Round-4 experiment record: compare what two forward kernels save for the backward (hs, cs, gates) on the
same inputs and time them in one process.  `FHVAE_FWD_WR_LOCKSTEP` was the A/B switch of the layer-halves kernel, which is not in
the tree: against the current library both legs run the same kernel (all differences 0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-scalablefhvae_amd"))
import torch
import hip_binding as hb

B, T, I, Ic, H, L = 2048, 20, 80, 0, 256, 2
torch.manual_seed(0)
lstm = torch.nn.LSTM(I + Ic, H, L)
names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
x = torch.randn(T, B, I).cuda()
res = {}
for mode in ("new", "lockstep"):
    if mode == "lockstep":
        os.environ["FHVAE_FWD_WR_LOCKSTEP"] = "1"
    hs_top, hn = hb.lstm_seq(x, None, T, params, hb.BF16)
    torch.cuda.synchronize()
    sv = hs_top.grad_fn.saved_tensors
    res[mode] = dict(hs=sv[2].float().clone(), cs=sv[3].clone(), gates=sv[4].float().clone(), top=hs_top.detach().clone(), hn=hn.detach().clone())
for k in ("top", "hn", "hs", "cs", "gates"):
    a, b = res["new"][k], res["lockstep"][k]
    d = (a - b).abs()
    print(k, tuple(a.shape), "max diff %.4g" % d.max().item(), "frac differing %.4g" % (d > 1e-6).float().mean().item())
    if d.max().item() > 1e-3:
        idx = (d > 1e-3).nonzero()
        print("   first bad index", idx[0].tolist(), "last", idx[-1].tolist(), "count", idx.shape[0])
        for dim in range(a.dim()):
            u = idx[:, dim].unique()
            print("   dim", dim, "distinct", u.numel(), "min", u.min().item(), "max", u.max().item(), (u[:20].tolist() if u.numel() <= 64 else ""))
a, b = res["new"]["gates"], res["lockstep"]["gates"]
d = (a - b).abs() > 1e-3
idx = d.nonzero()
torch.set_printoptions(precision=4, linewidth=200, sci_mode=False)
for n in range(0, min(idx.shape[0], 40000), 9973):
    l, t, row, col = idx[n].tolist()
    m = col // 128
    print("bad", l, t, row, col, "new", a[l, t, row, col].item(), "want", b[l, t, row, col].item())
    print("  new  row cols 88..127 of member:", a[l, t, row, m * 128 + 88:m * 128 + 128])
    print("  want row cols 88..127 of member:", b[l, t, row, m * 128 + 88:m * 128 + 128])
    v = a[l, t, row, col]
    for tt in (t - 1, t + 1):
        if 0 <= tt < T:
            print("   same slot at t%+d: want %.4f" % (tt - t, b[l, tt, row, col].item()), " layer0 same slot: %.4f" % b[0, tt, row, col].item())
    print("   layer0 same slot t: %.4f  t+1: %.4f t+2: %.4f" % (b[0, t, row, col].item(), b[0, min(t + 1, T - 1), row, col].item(), b[0, min(t + 2, T - 1), row, col].item()))
# per (t) count and per row-in-cluster histogram
print("bad per t:", d[1].sum(dim=(1, 2)).tolist())
rows = idx[:, 2] % 64
print("bad per row%64:", torch.bincount(rows, minlength=64).tolist())
print("bad per member:", torch.bincount(idx[:, 3] // 128, minlength=8).tolist())
print("bad per cluster (first 32):", torch.bincount(idx[:, 2] // 64, minlength=32).tolist())
# A/B timing of the whole forward call (operand casts + recurrence), same process
for mode in ("lockstep", "new", "lockstep", "new"):
    if mode == "lockstep":
        os.environ["FHVAE_FWD_WR_LOCKSTEP"] = "1"
    else:
        os.environ.pop("FHVAE_FWD_WR_LOCKSTEP", None)
    for _ in range(5):
        hb.lstm_seq(x, None, T, params, hb.BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        hb.lstm_seq(x, None, T, params, hb.BF16)
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f us per forward call" % (mode, e0.elapsed_time(e1) * 1000 / 50))
