"""K5 backward (bf16 compute mode; `f32` as first argument: the exact-f32 kernels), one-pass against two-pass (no workspace), per-kernel under rocprofv3:
rocprofv3 --kernel-trace -d /tmp/kd -o r -- python3 tools/bench_disc_lp.py && python3 tools/rocpd_stats.py <db> 1 20"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import torch, hip_binding as hb


LP = not (len(sys.argv) > 1 and sys.argv[1] == "f32")


def run(B, S, D=32, reps=5):
    q = torch.randn(B, D, device="cuda"); t = torch.randn(S, D, device="cuda"); idx = torch.randint(0, S, (B,), device="cuda")
    g = torch.ones(1, device="cuda")
    rmax, rsum, tgt, ce = hb.raw_disc_fwd(q, t, idx, lp=LP)
    for two in (False, True):
        fn = lambda: hb.raw_disc_bwd(q, t, idx, rmax, rsum, g, 1.0 / B, lp=LP, ws_bytes=0 if two else None)
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        print("%s B=%d S=%d %s: bwd %.3f ms" % ("bf16" if LP else "f32", B, S, "two-pass" if two else "one-pass", e0.elapsed_time(e1) / reps), flush=True)


for B, S in ((2048, 28000), (2048, 100000)) + (((2048, 1000000),) if not LP else ()):
    run(B, S)
