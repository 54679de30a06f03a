import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import torch, hip_binding as hb
def run(B, S, D=32):
    q = torch.randn(B, D, device="cuda"); t = torch.randn(S, D, device="cuda"); idx = torch.randint(0, S, (B,), device="cuda")
    g = torch.ones(1, device="cuda")
    def f():
        return hb.raw_disc_fwd(q, t, idx)
    rmax, rsum, tgt, ce = f()
    def b():
        return hb.raw_disc_bwd(q, t, idx, rmax, rsum, g, 1.0 / B)
    for name, fn in (("fwd", f), ("bwd", b)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%s B=%d S=%d: %s %.3f ms  (%.1f TFLOP/s algorithmic 3BSD%s)" % (os.environ.get("FHVAE_DISC_VALU") and "VALU" or "MFMA", B, S, name, ms, 3.0 * B * S * D * (1 if name == "fwd" else 3) / ms / 1e9, "" if name == "fwd" else " x3"))
run(256, 4600); run(2048, 28000); run(2048, 125000); run(2048, 1000000)
