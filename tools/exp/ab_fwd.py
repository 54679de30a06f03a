"""Same-box A/B of a forward-kernel switch (an environment variable the library reads per launch): alternate the two settings,
200 calls each, three rounds.  Record of DESIGN 9 item 5: the switches it was used with (FHVAE_FWD_PEEK_OFF, FHVAE_FWD_LATE_A,
FHVAE_FWD_TAIL8) existed only while the placements were compared; the library reads none of them now."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-scalablefhvae_amd"))
import torch
import hip_binding as hb
SW = sys.argv[1] if len(sys.argv) > 1 else "FHVAE_FWD_PEEK_OFF"
VAL = sys.argv[2] if len(sys.argv) > 2 else "1"
B, T, H, L = 2048, 20, 256, 2
for I, Ic in ((80, 0), (80, 32), (0, 64)):
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(I + Ic, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
    x = torch.randn(T, B, I).cuda() if I else None
    xc = torch.randn(B, Ic).cuda() if Ic else None
    res = {0: [], 1: []}
    for rnd in range(3):
        for off in (1, 0):
            if off:
                os.environ[SW] = VAL
            else:
                os.environ.pop(SW, None)
            for _ in range(10):
                hb.lstm_seq(x, xc, T, params, hb.BF16)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                hb.lstm_seq(x, xc, T, params, hb.BF16)
            e1.record(); torch.cuda.synchronize()
            res[off].append(e0.elapsed_time(e1) * 1000 / 200)
    print("I=%d Ic=%d: %s set %s us | unset %s us per forward call" % (I, Ic, SW, ["%.1f" % v for v in res[1]], ["%.1f" % v for v in res[0]]), flush=True)
assert hb.lstm_sync_status() == 0
