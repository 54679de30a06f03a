import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-scalablefhvae_amd"))
import torch, hip_binding as hb
def run(B, H, T, dt):
    params = []
    for l in range(2):
        kin = 80 if l == 0 else H
        params += [torch.randn(4 * H, kin, device="cuda") * 0.05, torch.randn(4 * H, H, device="cuda") * 0.05,
                   torch.zeros(4 * H, device="cuda"), torch.zeros(4 * H, device="cuda")]
    x = torch.randn(T, B, 80, device="cuda")
    for _ in range(5):
        hb.lstm_seq(x, None, T, params, dt)
    torch.cuda.synchronize()
run(16, 16, 3, hb.BF16)
run(256, 256, 3, hb.F32)
run(2048, 256, 3, hb.BF16)
