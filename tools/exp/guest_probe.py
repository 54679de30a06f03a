"""Round-4 experiment: does a small-register weight-gradient kernel (wgrad_guest_kernel) run beside the persistent backward
recurrence, and does the order in which the two reach the GPU matter?  rocprofv3 --kernel-trace ... then tools/rocpd_timeline.py.
Record of DESIGN 9 item 11: `FHVAE_WGRAD_GUEST` selected the guest kernel, which is not in the tree (against the current library the
side stream runs the 256x256 kernel, which cannot share a CU with the recurrence)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-scalablefhvae_amd"))
import torch
import hip_binding as hb

B, T, I, H, L = 2048, 20, 80, 256, 2
torch.manual_seed(0)
lstm = torch.nn.LSTM(I, H, L)
names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
x = torch.randn(T, B, I).cuda()
K = T * B
a = torch.randn(K, 1024, device="cuda").bfloat16()
b = torch.randn(K, 256, device="cuda").bfloat16()
c = torch.zeros(1024, 256, device="cuda")
side = torch.cuda.Stream()
marker = torch.zeros(1, device="cuda")


def guest(n=3):
    os.environ["FHVAE_WGRAD_GUEST"] = "1"
    for _ in range(n):
        hb.wgrad_bf16_(c, a, b)
    del os.environ["FHVAE_WGRAD_GUEST"]


mode = sys.argv[1] if len(sys.argv) > 1 else "guest_first"
for rep in range(4):
    hs, hn = hb.lstm_seq(x, None, T, params, hb.BF16)
    g = torch.ones_like(hs)
    torch.cuda.synchronize()
    marker.add_(1)  # (anchor kernel for the timeline tool)
    main = torch.cuda.current_stream()
    if mode == "guest_first":
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            guest(12)
        hs.backward(g)  # (reaches the GPU ~280 us later: eager launch latency)
    elif mode == "rec_first":
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            torch.cuda._sleep(750000)  # ~320 us: the first recurrence is running by then
            guest(8)
        hs.backward(g)
    else:
        hs.backward(g)
    main.wait_stream(side)
    hb.flush_param_grads()
    torch.cuda.synchronize()
print("done", mode)
