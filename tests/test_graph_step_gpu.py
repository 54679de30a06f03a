"""A training step captured into a hipGraph and replayed (bench.py's headline form, train_model.py --hip-graph) against the same
steps run eagerly: the captured step contains FusedAdam's launch that counts its own step and clears the gradient arena
(ABI 11), the cached backward seed, the deferred grouped weight gradients -- none of which the eager parity tests see through a
graph.  Reference loop body: train_model.py:446-454."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_ops_gpu import close, hb  # noqa: F401


def _build(dtype, H, B, seed=0):
    from fhvae import FHVAE
    from hip_optim import FusedAdam

    T, F, D, S = 20, 80, 32, 512
    torch.manual_seed(seed)
    model = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T, num_seqs=S, reference_compat=False, compute_dtype=dtype).cuda()
    opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.95, 0.999))
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, F, generator=g).cuda()
    idx = torch.randint(0, S, (B,), generator=g).cuda()
    ns = torch.full((B,), 7, dtype=torch.int64).cuda()
    return model, opt, (x, idx, ns), S


def _step(hb, model, opt, batch, S):
    from train_model import loss_function

    x, idx, ns = batch
    opt.zero_grad()
    out = model(x, idx, S, ns, eps=_step.eps)
    loss = loss_function(out[0], out[1], 10.0)
    hb.backward(loss)
    opt.step()
    return loss.detach()


@pytest.mark.parametrize("dtype,H,B", [("f32", 32, 64), ("bf16", 256, 256)])
def test_captured_step_replays_like_eager_steps(hb, dtype, H, B):
    n = 4
    # fixed reparameterisation draws: the captured step would otherwise replay its generator state differently from eager calls
    ge = torch.Generator().manual_seed(5)
    _step.eps = (torch.randn(B, 32, generator=ge).cuda(), torch.randn(B, 32, generator=ge).cuda())
    # ---- eager
    model, opt, batch, S = _build(dtype, H, B)
    losses_e = [float(_step(hb, model, opt, batch, S)) for _ in range(n)]
    torch.cuda.synchronize()
    p_e = opt.p_arena.flat.clone()
    assert int(opt.step_dev.item()) == n and float(opt.g_arena.flat.abs().max()) == 0.0
    # ---- captured: warm-up and capture must not train (state put back afterwards, like train_model.py's graph_step)
    model, opt, batch, S = _build(dtype, H, B)
    keep = [t.clone() for t in (opt.p_arena.flat, opt.m, opt.v, opt._step_buf)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            _step(hb, model, opt, batch, S)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_loss = _step(hb, model, opt, batch, S)
    for t, k in zip((opt.p_arena.flat, opt.m, opt.v, opt._step_buf), keep):
        t.copy_(k)
    losses_g = []
    for _ in range(n):
        graph.replay()
        losses_g.append(float(g_loss))
    torch.cuda.synchronize()
    assert hb.lstm_sync_status() == 0
    assert int(opt.step_dev.item()) == n                       # the Adam launch advanced the device-side count on every replay
    assert float(opt.g_arena.flat.abs().max()) == 0.0           # ... and left the gradient arena clear for the next one
    tol = 1e-5 if dtype == "f32" else 2e-3                      # (split-K float atomics: the summation order differs run to run)
    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) <= tol * max(1.0, abs(a)), (losses_e, losses_g)
    close(opt.p_arena.flat, p_e, rtol=tol, what="parameters after %d replays vs %d eager steps" % (n, n))
