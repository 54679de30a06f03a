"""Achieved HBM GB/s of the streaming kernels K3 (ELBO), K4 (gather), Adam, segment_gather at large sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import torch, hip_binding as hb
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
B, T, F, D, S = 65536, 20, 80, 32, 1000000
x = torch.randn(T, B, F, device="cuda"); xm = torch.randn(T, B, F, device="cuda"); xl = torch.randn(T, B, F, device="cuda") * .3
z = [torch.randn(B, D, device="cuda") for _ in range(5)]
ns = torch.randint(20, 200, (B,), device="cuda")
lay = (B, T, F, (F, B * F), (F, B * F))
s = t(lambda: hb.elbo(x, xm, xl, *z, ns, lay, True))
byt = B * (3 * T * F * 4 + 6 * D * 4 + 5 * 4)
print("K3 elbo fwd  B=%d: %.3f ms  %.0f GB/s (algorithmic %d B/segment)" % (B, s * 1e3, byt / s / 1e9, byt // B))
xm.requires_grad_(True); xl.requires_grad_(True); zz = [v.clone().requires_grad_(True) for v in z]
def fb():
    o = hb.elbo(x, xm, xl, *zz, ns, lay, False); o[0].sum().backward()
s2 = t(fb, 5)
print("K3 elbo fwd+bwd (incl. torch sum/ones): %.3f ms  ~%.0f GB/s on 3+3+2 planes" % (s2 * 1e3, B * 8 * T * F * 4 / s2 / 1e9))
table = torch.randn(S, D, device="cuda"); idx = torch.randint(0, S, (B,), device="cuda")
s = t(lambda: hb.raw_gather_rows(table, idx))
print("K4 gather    B=%d S=%d: %.3f ms  %.0f GB/s" % (B, S, s * 1e3, B * (2 * D * 4 + 8) / s / 1e9))
n = 64 * 1024 * 1024
p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4)); v.abs_(); step = torch.ones((), dtype=torch.int32, device="cuda")
s = t(lambda: hb.adam_step_(p, g, m, v, step, 1e-3, 0.95, 0.999, 1e-8))
print("Adam n=%d: %.3f ms  %.0f GB/s (7 x 4 B per parameter)" % (n, s * 1e3, n * 28 / s / 1e9))
pool = torch.randn(4_000_000, F, device="cuda"); st = torch.randint(0, 4_000_000 - T, (B,), device="cuda")
mean = torch.zeros(F, device="cuda"); istd = torch.ones(F, device="cuda")
s = t(lambda: hb.segment_gather(pool, st, T, mean, istd))
print("segment_gather B=%d: %.3f ms  %.0f GB/s" % (B, s * 1e3, 2 * B * T * F * 4 / s / 1e9))
