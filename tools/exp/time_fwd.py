"""Time the forward recurrence call alone (c3 net shapes): eager, per call = operand casts + the recurrence launch.
Used for DESIGN 9 item 5 (e.g. the library built with the saved-for-backward stores compiled out: 184 us against 209)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-scalablefhvae_amd"))
import torch
import hip_binding as hb
B, T, H, L = 2048, 20, 256, 2
for I, Ic in ((80, 0), (80, 32), (0, 64)):
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(I + Ic, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
    x = torch.randn(T, B, I).cuda() if I else None
    xc = torch.randn(B, Ic).cuda() if Ic else None
    for _ in range(5):
        hb.lstm_seq(x, xc, T, params, hb.BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        hb.lstm_seq(x, xc, T, params, hb.BF16)
    e1.record(); torch.cuda.synchronize()
    print("I=%d Ic=%d: %.1f us per forward call (casts + recurrence)" % (I, Ic, e0.elapsed_time(e1) * 1000 / 50), flush=True)
