"""Phase clocks of the register-stationary forward recurrence (csrc/lstm_fwd_wr.hip), cluster 0 / member 0: FHVAE_CLUSTER_TLOG=1
makes the kernel log wall_clock64() (100 MHz) per step at: 0 step begins, 1 images landed (P1), 2 layer 0's products done and the
h1 image requested (P2 + P3), 3 layer 0's gate math done, 4 h0 out + A published, 5 layer 1's products of h0 done + barrier (P5, P6),
6 layer 0's saved-for-backward stores issued (P7), 7 layer 1's recurrent products, the h0 request, gate math, h1 out, B published."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-scalablefhvae_amd"))
os.environ["FHVAE_CLUSTER_TLOG"] = "1"
import torch
import hip_binding as hb

H, L, T = 256, 2, 20
for B, I, Ic in [(2048, 80, 0), (2048, 80, 32), (2048, 0, 64)]:
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(I + Ic, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    params = [getattr(lstm, n).detach().cuda() for n in names]
    x = torch.randn(T, B, I).cuda() if I else None
    xc = torch.randn(B, Ic).cuda() if Ic else None
    for rep in range(3):
        hs, hn = hb.lstm_seq(x, xc, T, params, hb.BF16)
    torch.cuda.synchronize()
    lp = hb.LSTM_WORKSPACES[-1]
    n = T + 1
    log = lp[12288:12288 + 8 * 8 * n].view(torch.int64).cpu().view(n, 8).double() * 0.01  # us
    print("B=%d I=%d Ic=%d: %.1f us from the first step to the last; per step: P1 image wait | P2+P3 tail stores, acc0 | P4 L0 cells | h0 out + publish A | P5 W_ih1 + P6 barrier | P7 poll A + requests (wave 0) | P8 W_hh1, L1 cells, h1 out, publish B" % (B, I, Ic, (log[-1, 7] - log[0, 0]).item()))
    for s in range(n):
        a = log[s]
        print("  s=%2d  %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f   = %5.2f" % (s, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[6] - a[5], a[7] - a[6], a[7] - a[0]))
