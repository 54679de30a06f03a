"""Where the 512 workgroups of the persistent H=512 forward land: HW_REG_HW_ID per (xcd, ticket slot)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-scalablefhvae_amd"))
os.environ["FHVAE_CLUSTER_TLOG"] = "1"
os.environ["FHVAE_STREAM_STAGGER_F"] = "-1"
import torch
import hip_binding as hb
H, L, T, I, Ic, B = 512, 2, 4, 80, 32, 2048
torch.manual_seed(0)
lstm = torch.nn.LSTM(I + Ic, H, L)
names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
params = [getattr(lstm, n).detach().cuda() for n in names]
x = torch.randn(T, B, I).cuda(); xc = torch.randn(B, Ic).cuda()
for rep in range(2):
    hs, hn = hb.lstm_seq(x, xc, T, params, hb.BF16)
torch.cuda.synchronize()
lp = hb.LSTM_WORKSPACES[-1]
hw = lp[12288:12288 + 2048].view(torch.int32).cpu().view(8, 64)
for x_ in range(2):
    print("xcd", x_)
    for s in range(64):
        v = int(hw[x_, s]) & 0xffffffff
        print("  slot %2d hw %08x  wave %d simd %d pipe %d cu %2d sh %d se %d" % (s, v, v & 15, (v >> 4) & 3, (v >> 6) & 3, (v >> 8) & 15, (v >> 12) & 1, (v >> 13) & 7))
