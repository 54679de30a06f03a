"""Micro-benchmarks of individual C-ABI ops (HIP events, back-to-back launches)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import torch
import hip_binding as hb

lib = hb.load_library()

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

def gemm(M, K, N, dtype):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
    y = torch.empty(M, N, device="cuda")
    if dtype == hb.BF16:
        x, w = x.bfloat16(), w.bfloat16()
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.fhvae_linear_fwd(x.data_ptr(), K, w.data_ptr(), K, None, y.data_ptr(), N, None, M, K, N, 0, dtype, st)
    us = timeit(f)
    print("gemm %s M=%d K=%d N=%d: %.1f us  %.1f TFLOP/s" % ("bf16" if dtype else "f32", M, K, N, us, 2.0 * M * K * N / us / 1e6))

def wgrad(K, M, N, n=1):
    """C[M,N] += A[K,M]^T . B[K,N], bf16 K-major operands (csrc/wgrad.hip): the weight-gradient contraction; n problems at once
    through fhvae_wgrad_bf16 one by one (the model's grouped launch is timed by bench.py's op timers)."""
    a = torch.randn(K, M, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16()
    c = torch.zeros(M, N, device="cuda")
    us = timeit(lambda: hb.wgrad_bf16_(c, a, b))
    print("wgrad bf16 K=%d M=%d N=%d: %.1f us  %.1f TFLOP/s" % (K, M, N, us, 2.0 * M * K * N / us / 1e6))


def lstm(B, T, I, Ic, H, L, dtype, bwd=True):
    torch.manual_seed(0)
    params = []
    for l in range(L):
        kin = I + Ic if l == 0 else H
        params += [torch.randn(4 * H, kin, device="cuda") * 0.05, torch.randn(4 * H, H, device="cuda") * 0.05,
                   torch.zeros(4 * H, device="cuda"), torch.zeros(4 * H, device="cuda")]
    params = [p.requires_grad_(True) for p in params]
    x = torch.randn(T, B, I, device="cuda") if I else None
    xc = torch.randn(B, Ic, device="cuda", requires_grad=True) if Ic else None
    hb.OP_TIMER.enable()
    for _ in range(8):
        hs, hn = hb.lstm_seq(x, xc, T, params, dtype)
        if bwd:
            (hs.sum() + hn.sum()).backward()
    s = hb.OP_TIMER.summary(); hb.OP_TIMER.disable()
    fl = 2.0 * T * B * sum(4 * H * ((I + Ic if l == 0 else H) + H) for l in range(L))
    f = s["fhvae_lstm_seq_fwd"]; 
    msg = "lstm %s B=%d I=%d Ic=%d H=%d: fwd %.0f us (%.1f TF)" % ("bf16" if dtype else "f32", B, I, Ic, H, f[1] / f[0] * 1e3, fl / (f[1] / f[0] * 1e-3) / 1e12)
    if bwd:
        b = s["fhvae_lstm_seq_bwd"]
        msg += "  bwd %.0f us (%.1f TF)" % (b[1] / b[0] * 1e3, 2 * fl / (b[1] / b[0] * 1e-3) / 1e12)
    print(msg)

if __name__ == "__main__":
    x = torch.randn(4, 4, 4, device="cuda")
    print("tiny kernel launch (to_time_major 64 elems): %.1f us" % timeit(lambda: hb.to_time_major(x), 200))
    st = torch.cuda.current_stream().cuda_stream
    o = torch.empty(64, device="cuda")
    print("tiny raw launch: %.1f us" % timeit(lambda: lib.fhvae_to_time_major(x.data_ptr(), o.data_ptr(), None, 4, 4, 4, 0, st), 500))
    for dt in (hb.F32, hb.BF16):
        gemm(4096, 4096, 4096, dt)
        gemm(5120, 80, 1024, dt)
        gemm(256, 256, 1024, dt)
        gemm(256, 512, 1024, dt)
        gemm(2048, 512, 1024, dt)
    wgrad(4096, 4096, 4096)
    wgrad(40960, 1024, 256)
    wgrad(40960, 1024, 80)
    wgrad(5120, 1024, 256)
    wgrad(40960, 2048, 512)
    if "--gemm-only" in sys.argv:
        sys.exit(0)
    for dt in (hb.F32, hb.BF16):
        lstm(256, 20, 80, 0, 256, 2, dt)
        lstm(256, 20, 80, 32, 256, 2, dt)
        lstm(256, 20, 0, 64, 256, 2, dt)
        lstm(2048, 20, 80, 0, 256, 2, dt)
