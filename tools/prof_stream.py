"""Phase clocks of the persistent H=512 cells (lstm_stream.hip; row tile 0, unit tile 0, both layers): FHVAE_CLUSTER_TLOG=1 makes
the kernels log wall_clock64() (100 MHz) at wait-begin / wait-end / contraction-end / stores-issued / published per step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-scalablefhvae_amd"))
os.environ["FHVAE_CLUSTER_TLOG"] = "1"
os.environ["FHVAE_STREAM"] = "1"      # the persistent form is opt-in
os.environ["FHVAE_BIG_CELLS"] = "1"   # (also below the batch the heuristic takes)
import torch
import hip_binding as hb

H, L, T, I, Ic = 512, 2, 20, 80, 32
B = int(os.environ.get("PROF_B", "2048"))
torch.manual_seed(0)
lstm = torch.nn.LSTM(I + Ic, H, L)
names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
x = torch.randn(T, B, I).cuda()
xc = torch.randn(B, Ic).cuda().requires_grad_(True)
bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
for rep in range(3):
    hs, hn = hb.lstm_seq(x, xc, T, params, hb.BF16)
    if bwd:
        (hs.sum() + hn.sum()).backward()
torch.cuda.synchronize()
assert hb.lstm_sync_status() == 0
lp = hb.LSTM_WORKSPACES[-1]
for l in range(L):
    log = lp[12288 + l * 2048:12288 + l * 2048 + 8 * 8 * T].view(torch.int64).cpu().view(T, 8)[:, :5].double() * 0.01  # us
    print("%s layer %d: total %.1f us; per step: wait, contraction, epilogue (stores issued), publish, [gap to next step]" % (
        "bwd" if bwd else "fwd", l, (log[-1, 4] - log[0, 0]).item()))
    for s in range(T):
        nxt = log[s + 1, 0] if s + 1 < T else log[s, 4]
        a = log[s]
        print("  s=%2d  %6.2f %6.2f %6.2f %6.2f %6.2f   (step %.2f)" % (s, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], nxt - a[4], nxt - a[0]))
