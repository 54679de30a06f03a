"""Same-box A/B of a backward-kernel switch (an environment variable the library reads per launch): forward once, then the
backward of the same graph 100 times per setting (retain_graph), alternating, three rounds."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-scalablefhvae_amd"))
import torch
import hip_binding as hb
SW = sys.argv[1] if len(sys.argv) > 1 else "FHVAE_BWD_HOIST"
B, T, H, L = 2048, 20, 256, 2
for I, Ic in ((80, 0), (0, 64)):
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(I + Ic, H, L)
    names = [n + "_l%d" % l for l in range(L) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    params = [getattr(lstm, n).detach().cuda().requires_grad_(True) for n in names]
    x = torch.randn(T, B, I).cuda() if I else None
    xc = torch.randn(B, Ic).cuda().requires_grad_(True) if Ic else None
    hs, hn = hb.lstm_seq(x, xc, T, params, hb.BF16)
    g = torch.randn_like(hs)
    res = {0: [], 1: []}
    for rnd in range(3):
        for on in (0, 1):
            if on:
                os.environ[SW] = "1"
            else:
                os.environ.pop(SW, None)
            for _ in range(5):
                hs.backward(g, retain_graph=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                hs.backward(g, retain_graph=True)
            e1.record(); torch.cuda.synchronize()
            res[on].append(e0.elapsed_time(e1) * 1000 / 100)
    print("I=%d Ic=%d: %s unset %s us | set %s us per backward call (2 recurrences + projection + weight gradients)"
          % (I, Ic, SW, ["%.1f" % v for v in res[0]], ["%.1f" % v for v in res[1]]), flush=True)
assert hb.lstm_sync_status() == 0
