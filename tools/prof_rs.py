"""Phase clocks of the partial-dh per-layer backward (csrc/lstm_bwd_rs.hip), cluster 0 / member 0: FHVAE_CLUSTER_TLOG=1 makes
the kernel log wall_clock64() (100 MHz) per step at: 0 step begins, 2 flags seen, 3 partials added + epilogue + image barrier,
4 MFMAs + partial stores issued, 5 published.  The lower layer's launch logs at slot 0, the top layer's at slot 256."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-scalablefhvae_amd"))
os.environ["FHVAE_CLUSTER_TLOG"] = "1"
import torch
import hip_binding as hb

H, L, T, I = 256, 2, 20, 80
for B in [int(b) for b in os.environ.get("PROF_B", "2048").split(",")]:
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(I, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
    x = torch.randn(T, B, I).cuda()
    for rep in range(3):
        hs, hn = hb.lstm_seq(x, None, T, params, hb.BF16)
        (hs.sum() + hn.sum()).backward()
    torch.cuda.synchronize()
    lp = hb.LSTM_WORKSPACES[-1]
    for name, off in (("top layer", 256), ("lower layer", 0)):
        log = lp[12288 + off * 8:12288 + off * 8 + 8 * 8 * T].view(torch.int64).cpu().view(T, 8).double() * 0.01  # us
        print("    after join -> weights and first operands requested %.2f us; last publish -> bias sums added %.2f us" % ((log[0, 7] - log[0, 6]).item(), (log[T - 1, 7] - log[T - 1, 5]).item()))
        print("B=%d %s: recurrence %.1f us; per step: wait | loads+epilogue | mfma+stores | publish | tail" % (B, name, (log[-1, 5] - log[0, 0]).item()))
        for s in range(T):
            a = log[s]
            nxt = log[s + 1, 0] if s + 1 < T else a[5]
            w = (a[2] - a[0]) if s > 0 else 0.0
            e0 = a[2] if s > 0 else a[0]
            print("  s=%2d  %5.2f %5.2f %5.2f %5.2f %5.2f" % (s, w, a[3] - e0, a[4] - a[3], a[5] - a[4], nxt - a[5]))
