"""Which runtime memset / memcpy launches the one-rank distributed step issues, with the Python frames that cause them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", WORLD_SIZE="1")
import torch
import torch.distributed as dist
from torch.profiler import profile, ProfilerActivity
import bench
from fhvae import FHVAE
from dist_shard import DistributedFHVAE

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
cfg = bench.CONFIGS["c3"]
H, L, D, S, T, F = (cfg[k] for k in "HLDSTF")
torch.manual_seed(0)
model = FHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, seg_len=T, num_seqs=S, reference_compat=False, compute_dtype="bf16").cuda()
runner = DistributedFHVAE(model, lr=1e-3, betas=(0.95, 0.999))
x, idx, ns = (t.cuda() for t in bench.synth_cpu(cfg, 2048, 0, "uniform"))
for _ in range(3):
    runner.train_step(x, idx, ns, alpha=10.0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    runner.train_step(x, idx, ns, alpha=10.0)
    torch.cuda.synchronize()
cnt = {}
for ev in prof.events():
    n = ev.name
    if "emset" in n or "emcpy" in n or n.startswith("aten::"):
        st = [f for f in (ev.stack or []) if "pytorch-scalablefhvae_amd" in f][:2]
        k = (n, tuple(st))
        cnt[k] = cnt.get(k, 0) + 1
for (n, st), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print("%3d x %-28s %s" % (c, n, " <- ".join(s.split("/")[-1] for s in st)))
dist.destroy_process_group()
