import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-scalablefhvae_amd"))
import torch
import hip_binding as hb
from fhvae import FHVAE
from hip_optim import FusedAdam

def run(cluster, B, steps=4):
    if cluster: os.environ.pop("FHVAE_NO_CLUSTER", None)
    else: os.environ["FHVAE_NO_CLUSTER"] = "1"
    torch.manual_seed(0)
    H, L, D, S, T, F = 256, 2, 32, 4600, 20, 80
    dev = torch.device("cuda:0")
    model = FHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, seg_len=T, num_seqs=S, reference_compat=False, compute_dtype="bf16").to(dev)
    opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.95, 0.999))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, F, generator=g).to(dev)
    idx = torch.randint(0, S, (B,), generator=g).to(dev)
    nsegs = torch.randint(20, 200, (B,), generator=g).to(dev)
    out = []
    for i in range(steps):
        opt.zero_grad()
        lb, dl, lpx, k1, k2, lpm = model(x, idx, S, nsegs)
        loss = -(lb + 10.0 * dl).mean()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        out.append((loss.item(), lb.mean().item(), hb.lstm_sync_status()))
    return out

for B in (256, 2048):
    a = run(True, B); b = run(False, B)
    for u, v in zip(a, b):
        print(B, "cluster", u, "step", v)
