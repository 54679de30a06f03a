import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import microbench as mb, hip_binding as hb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = hb.BF16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else hb.F32
mb.lstm(B, 20, 80, 32, 256, 2, dt)
