"""One forward + backward of the H=512 bf16 LSTM net at the c4 shape, timed; prints progress lines (hang probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-scalablefhvae_amd")]
import torch
import hip_binding as hb
hb.load_library()
B, T, I, Ic, H, L = int(os.environ.get("PB", 2048)), 20, 80, 32, int(os.environ.get("PH", 512)), 2
torch.manual_seed(0)
lstm = torch.nn.LSTM(I + Ic, H, L, batch_first=True)
names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
x = torch.randn(T, B, I).cuda(); xc = torch.randn(B, Ic).cuda().requires_grad_(True)
print("inputs ready", flush=True)
for it in range(3):
    t0 = time.time()
    hs, hn = hb.lstm_seq(x, xc, T, params, hb.BF16)
    torch.cuda.synchronize(); t1 = time.time()
    print("fwd %d: %.1f ms (form %d)" % (it, (t1 - t0) * 1e3, hb.LAST_LSTM_FORM["form"]), flush=True)
    (hs.sum() + hn.sum()).backward()
    torch.cuda.synchronize()
    print("bwd %d: %.1f ms" % (it, (time.time() - t1) * 1e3), flush=True)
print("status", hb.lstm_sync_status(), "grad finite", all(torch.isfinite(p.grad).all().item() for p in params), flush=True)
