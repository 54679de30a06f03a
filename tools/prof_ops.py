"""Which torch ops launch the small fill / copy kernels of one training step (torch.profiler, eager step).
usage: python tools/prof_ops.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import torch
from torch.profiler import ProfilerActivity, profile

import hip_binding as hb
from fhvae import FHVAE
from hip_optim import FusedAdam
from train_model import loss_function

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, F, H, D, S = 20, 80, 256, 32, 4600
dev = torch.device("cuda:0")
hb.load_library()
torch.manual_seed(0)
model = FHVAE(T * F, [H] * 2, [H] * 2, D, D, [H] * 2, seg_len=T, num_seqs=S, reference_compat=False, compute_dtype="bf16").to(dev)
opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.95, 0.999))
x = torch.randn(B, T, F, device=dev)
idx = torch.randint(0, S, (B,), device=dev)
ns = torch.randint(20, 200, (B,), device=dev)


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
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.kernels:
        if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::") and ev.cpu_parent.kernels:
            continue
        st = [s for s in (ev.stack or []) if "site-packages" not in s and "dist-packages" not in s and "<built-in" not in s]
        rows.append((ev.name, [k.name[:40] for k in ev.kernels], st[:3]))
for r in rows:
    print(r)
print(len(rows), "aten ops with kernels")
for ev in prof.events():
    if "emcpy" in ev.name or "emset" in ev.name:
        print(ev.name, ev.device_type, ev.cuda_time if hasattr(ev, "cuda_time") else "")
