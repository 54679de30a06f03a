"""Achieved HBM GB/s of the streaming kernels K3 (lower bound), K4 (gather), Adam, segment_gather at large sizes: the table
behind bench.py's roofline.hbm record (same function).  Under rocprofv3 --kernel-trace the per-kernel durations of the same
launches land in the trace (tools/collect_r03.sh writes both into profiles/r03_stream_kernels_hbm.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import torch
import bench

rec = bench.hbm_record(torch.device("cuda", 0))
print("kernel            launch_us   algorithmic_MB     GB/s   of 8 TB/s   (HIP events on the launch stream, 10 launches)")
for k, v in rec.items():
    print("%-16s %10.1f %16.1f %8.0f %10.3f   %s" % (k, v["us"], v["bytes"] / 1e6, v["gbps"], v["frac"], v.get("note", "")))
