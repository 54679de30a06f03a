"""Phase clocks of the persistent LSTM recurrence (cluster 0, member 0): FHVAE_CLUSTER_TLOG=1 makes the kernel log
wall_clock64() (100 MHz) at wait-begin / wait-end / contraction-end / h-stored / published per step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-scalablefhvae_amd"))
os.environ["FHVAE_CLUSTER_TLOG"] = "1"
import torch
import hip_binding as hb

H, L, T, I = 256, 2, 20, 80
for B in [int(b) for b in os.environ.get("PROF_B", "256,2048").split(",")]:
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(I, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    params = [getattr(lstm, n).detach().cuda() for n in names]
    x = torch.randn(T, B, I).cuda()
    bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
    params = [q.requires_grad_(True) for q in params]
    for rep in range(3):
        hs, hn = hb.lstm_seq(x, None, T, params, hb.BF16)
        if bwd:
            (hs.sum() + hn.sum()).backward()
    torch.cuda.synchronize()
    lp = hb.LSTM_WORKSPACES[-1]
    log = lp[12288:12288 + 8 * 8 * (T + L - 1)].view(torch.int64).cpu().view(T + L - 1, 8)[:, :5].double() * 0.01  # us
    t0 = log[0, 0]
    print("B=%d  total %.1f us; per step: wait, contraction, gates+h store, publish, [tail stores until next step]" % (B, (log[-1, 4] - t0).item()))
    for s in range(T + L - 1):
        nxt = log[s + 1, 0] if s + 1 < T + L - 1 else log[s, 4]
        a = log[s]
        print("  s=%2d  %5.2f %5.2f %5.2f %5.2f %5.2f" % (s, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], nxt - a[4]))
