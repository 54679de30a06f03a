"""Times fhvae_proj_bf16 (csrc/proj.hip) and, beside it, the generic engine on the same shape (HIP events, back-to-back)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-scalablefhvae_amd"))
import torch
import hip_binding as hb

lib = hb.load_library()

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, K in [(40960, 256, 1024), (40960, 160, 256), (40960, 1024, 256), (81920, 512, 2048), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda")
    us = timeit(lambda: hb.proj_bf16(a, w, out=out))
    st = torch.cuda.current_stream().cuda_stream
    us_g = timeit(lambda: lib.fhvae_linear_fwd(a.data_ptr(), K, w.data_ptr(), K, None, out.data_ptr(), N, None, M, K, N, 0, hb.BF16, st))
    print("proj M=%d N=%d K=%d: %.1f us %.0f TFLOP/s   (generic engine %.1f us %.0f TFLOP/s)" % (M, N, K, us, 2.0 * M * N * K / us / 1e6, us_g, 2.0 * M * N * K / us_g / 1e6))
