"""Which ATen / runtime kernels still run inside a training step (c3 shape), with the Python frames that launch them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from fhvae import FHVAE
from hip_optim import FusedAdam
from train_model import loss_function

cfg = bench.CONFIGS["c3"]
H, L, D, S, T, F = (cfg[k] for k in "HLDSTF")
torch.manual_seed(0)
model = FHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, seg_len=T, num_seqs=S, reference_compat=False, compute_dtype="bf16").cuda()
opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.95, 0.999))
x, idx, ns = (t.cuda() for t in bench.synth_cpu(cfg, 2048, 0, "uniform"))

def step():
    opt.zero_grad()
    out = model(x, idx, S, ns)
    loss = loss_function(out[0], out[1], 10.0)
    loss.backward()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ka in prof.key_averages(group_by_stack_n=12):
    t = getattr(ka, "self_device_time_total", None)
    if t is None:
        t = getattr(ka, "self_cuda_time_total", 0.0)
    if (ka.key.startswith("aten::") or "Memcpy" in ka.key or "memcpy" in ka.key) and t > 0:
        st = [f for f in (ka.stack or []) if ("scalablefhvae" in f or "aten_trace" in f or "bench.py" in f)][:4] or list(ka.stack or [])[:6]
        rows.append((t, ka.key, ka.count, st))
for t, name, n, st in sorted(rows, key=lambda r: -r[0])[:30]:
    print("%-30s x%d  %.1f us" % (name, n, t))
    for f in st:
        print("      ", f)
