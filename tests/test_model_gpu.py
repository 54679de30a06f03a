"""End-to-end GPU parity of the two drop-in modules: SimpleFHVAE against the golden vectors taken
from the reference, FHVAE (LSTM) against the CPU oracle built on torch.nn.LSTM.  1e-4 relative fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from test_ops_gpu import close, dev, hb  # noqa: F401

OUT = ["lower_bound", "log_qy", "log_px_z", "neg_kld_z1", "neg_kld_z2", "log_pmu2"]


def test_simple_tiny_golden_forward_loss_grads(hb, golden_dir):
    from simple_fhvae import SimpleFHVAE

    g = dict(np.load(os.path.join(golden_dir, "simple_tiny_f32.npz")))
    m = SimpleFHVAE(4 * 8, [16, 16], [16, 16], 16, 16, [16, 16])
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")}, strict=False)
    m.cuda()
    table = dev(torch.from_numpy(g["table"])).requires_grad_(True)
    # idx and nsegs arrive as CPU tensors, like the reference loop builds them (train_model.py:445)
    out = m(dev(torch.from_numpy(g["x"])), torch.from_numpy(g["idx"]), 12, torch.from_numpy(g["nsegs"]), mu2_table=table,
            eps=(torch.from_numpy(g["eps_z2"]), torch.from_numpy(g["eps_z1"])))
    for k, n in enumerate(OUT):
        close(out[k], torch.from_numpy(g["out_" + n]), what=n)
    assert not out[2].requires_grad and not out[5].requires_grad
    from train_model import loss_function

    loss = loss_function(out[0], out[1], float(g["alpha"]))
    close(loss, torch.from_numpy(g["loss"]), what="loss")
    loss.backward()
    for n, p in m.named_parameters():
        if n == "mu2_table":
            continue
        ref = g["grad_" + n]
        if ref.size == 0:
            assert p.grad is None, n  # decoder gets no gradient in the reference
        else:
            close(p.grad, torch.from_numpy(ref), what="grad " + n)
    close(table.grad, torch.from_numpy(g["grad_table"]), what="grad table")
    assert m.qz2_x[0].shape == (8, 16) and m.pz2[0].shape == (8, 16)  # utils.py:52,58 attributes


def test_simple_refshape_golden_forward(hb, golden_dir):
    from simple_fhvae import SimpleFHVAE

    g = dict(np.load(os.path.join(golden_dir, "simple_refshape_f32.npz")))
    T, F, D, B, S = [int(v) for v in g["meta_TFDBS"]]
    m = SimpleFHVAE(T * F)
    R.fill_state_dict_det(m, seed=1.0)
    m.cuda()
    with torch.no_grad():
        out = m(dev(R.det_tensor((B, T, F), seed=3.0)), R.det_index(B, S, seed=5), S, R.det_index(B, 180, seed=9) + 20,
                mu2_table=dev(torch.from_numpy(g["table"])),
                eps=(torch.from_numpy(g["eps_z2"]), torch.from_numpy(g["eps_z1"])))
    for k, n in enumerate(OUT):
        close(out[k], torch.from_numpy(g["out_" + n]), what=n)


@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("cfg", [dict(T=4, F=6, H=8, D=4, B=5, S=9), dict(T=20, F=80, H=64, D=16, B=48, S=300),
                                 dict(T=20, F=80, H=256, D=32, B=64, S=500)])
def test_fhvae_lstm_vs_oracle(hb, cfg, compat):
    from fhvae import FHVAE
    from train_model import loss_function

    T, F, H, D, B, S = (cfg[k] for k in "TFHDBS")
    torch.manual_seed(H + B)
    ref = R.FHVAERef(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T)
    m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T, reference_compat=compat)
    m.load_state_dict(ref.state_dict(), strict=False)
    m.cuda()
    x = torch.randn(B, T, F)
    idx = torch.randint(0, S, (B,))
    idx[1] = idx[0]
    ns = torch.randint(3, 100, (B,))
    table = torch.randn(S, D, requires_grad=True)
    e2, e1 = torch.randn(B, D), torch.randn(B, D)
    want = ref(x, idx, S, ns, mu2_table=table, eps_z2=e2, eps_z1=e1, reference_compat=compat)
    R.loss_function(want[0], want[1], 10.0).backward()
    td = dev(table.detach()).requires_grad_(True)
    got = m(dev(x), idx, S, ns, mu2_table=td, eps=(e2, e1))
    for k, n in enumerate(OUT):
        close(got[k], want[k], what=n)
    loss_function(got[0], got[1], 10.0).backward()
    rp = dict(ref.named_parameters())
    for n, p in m.named_parameters():
        if n == "mu2_table":
            continue
        if rp[n].grad is None:
            assert p.grad is None, n
        else:
            # gradients flow through 20 recurrent steps: allow 5e-4 of the tensor's scale
            close(p.grad, rp[n].grad, rtol=5e-4, what="grad " + n)
    close(td.grad, table.grad, rtol=5e-4, what="grad table")


def test_persistent_table_trains(hb):
    """Default mode: persistent learnable mu2 table + on-device draws; a few Adam steps lower the loss."""
    from fhvae import FHVAE
    from train_model import loss_function

    torch.manual_seed(0)
    T, F, H, D, B, S = 20, 80, 32, 16, 64, 40
    m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], num_seqs=S, reference_compat=False).cuda()
    assert m.mu2_table.shape == (S, D) and m.mu2_table.is_cuda
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
    x = torch.randn(B, T, F).cuda()
    idx = torch.randint(0, S, (B,))
    losses = []
    for _ in range(8):
        opt.zero_grad()
        out = m(x, idx, S, 30)
        loss = loss_function(out[0], out[1], 10.0)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert m.mu2_table.grad is not None and m.pre_decoder.lstm.weight_hh_l1.grad.abs().sum() > 0


def test_fhvae_bf16_tracks_f32(hb):
    """bf16 MFMA operands (compute_dtype='bf16') against the f32 oracle: the bound and the loss agree to 1e-2
    relative ("matched ELBO" tolerance for the bf16 configs)."""
    from fhvae import FHVAE
    from train_model import loss_function

    T, F, H, D, B, S = 20, 80, 256, 32, 64, 500
    torch.manual_seed(5)
    ref = R.FHVAERef(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T)
    m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T, reference_compat=False, compute_dtype="bf16")
    m.load_state_dict(ref.state_dict(), strict=False)
    m.cuda()
    x, idx, ns = torch.randn(B, T, F), torch.randint(0, S, (B,)), torch.randint(3, 100, (B,))
    table, e2, e1 = torch.randn(S, D), torch.randn(B, D), torch.randn(B, D)
    with torch.no_grad():
        want = ref(x, idx, S, ns, mu2_table=table, eps_z2=e2, eps_z1=e1, reference_compat=False)
    td = dev(table).requires_grad_(True)
    got = m(dev(x), idx, S, ns, mu2_table=td, eps=(e2, e1))
    for k, n in enumerate(OUT):
        close(got[k], want[k], rtol=1e-2, what=n)
    loss = loss_function(got[0], got[1], 10.0)
    close(loss, R.loss_function(want[0], want[1], 10.0), rtol=1e-2, what="loss")
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for n, p in m.named_parameters() if n != "mu2_table")


@pytest.mark.parametrize("H,B,dtype,port", [(32, 48, "f32", 29617), (256, 64, "bf16", 29618), (256, 1024, "bf16", 29619)])
def test_distributed_wrapper_world1_matches_single_gpu(hb, H, B, dtype, port, monkeypatch):
    """dist_shard.DistributedFHVAE on a 1-rank RCCL group (the HipBackend code path of the sharded ops; FHVAE_DIST_NO_SOLO=1 so
    that the collectives are really issued to RCCL instead of being skipped as the identities they are on one rank):
    same losses as the plain single-GPU loop over several Adam steps.  The bf16 cases run the persistent LSTM kernels
    (contraction-split and rows form): there the runner reduces the first two gradient buckets from its hook between a
    net's recurrence and its parameter gradients."""
    import torch.distributed as dist
    from dist_shard import DistributedFHVAE
    from fhvae import FHVAE
    from hip_optim import FusedAdam
    from train_model import loss_function

    T, F, D, S = 20, 80, 16, 37
    x = torch.randn(B, T, F, generator=torch.Generator().manual_seed(1)).cuda()
    idx = torch.randint(0, S, (B,), generator=torch.Generator().manual_seed(2)).cuda()
    ns = torch.randint(20, 200, (B,), generator=torch.Generator().manual_seed(3)).cuda()
    eps = (torch.randn(B, D, generator=torch.Generator().manual_seed(4)).cuda(),
           torch.randn(B, D, generator=torch.Generator().manual_seed(5)).cuda())

    def build():
        torch.manual_seed(11)
        return FHVAE(T * F, [H, H], [H, H], D, D, [H, H], num_seqs=S, reference_compat=False, compute_dtype=dtype).cuda()

    m1 = build()
    opt = FusedAdam(m1.parameters(), lr=1e-3, betas=(0.95, 0.999))
    ref_losses = []
    for _ in range(3):
        opt.zero_grad()
        out = m1(x, idx, S, ns, eps=eps)
        loss = loss_function(out[0], out[1], 10.0)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())

    if port % 2:  # odd ports: real RCCL calls; even: the skipping default
        monkeypatch.setenv("FHVAE_DIST_NO_SOLO", "1")
    else:
        monkeypatch.delenv("FHVAE_DIST_NO_SOLO", raising=False)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        m2 = build()
        runner = DistributedFHVAE(m2, lr=1e-3, betas=(0.95, 0.999))
        assert runner.sh.solo == (not port % 2)
        fwd = m2.forward
        m2.forward = lambda *a, **k: fwd(*a, eps=eps, **k)  # same draws as the reference loop
        got = [runner.train_step(x, idx, ns, alpha=10.0)[0].item() for _ in range(3)]
    finally:
        dist.destroy_process_group()
    assert hb.lstm_sync_status() == 0
    tol = 1e-4 if dtype == "f32" else 2e-3  # bf16: the bias-gradient atomics sum in a different order run to run
    for a, b in zip(got, ref_losses):
        assert abs(a - b) <= tol * abs(b), (got, ref_losses)
    close(runner.shard, m1.mu2_table, rtol=tol, what="table after 3 steps")


def test_fused_adam_grad_sinks_match_torch_adam(hb):
    """FusedAdam (flat arena; backward kernels accumulate straight into it) against torch.optim.Adam driving
    the same HIP model through plain autograd gradients: same parameters after several steps."""
    from fhvae import FHVAE
    from hip_optim import FusedAdam
    from train_model import loss_function

    T, F, H, D, B, S = 20, 80, 32, 16, 40, 29
    x = torch.randn(B, T, F, generator=torch.Generator().manual_seed(1)).cuda()
    idx = torch.randint(0, S, (B,), generator=torch.Generator().manual_seed(2))
    idx[3] = idx[2]
    ns = torch.randint(20, 200, (B,), generator=torch.Generator().manual_seed(3))
    eps = (torch.randn(B, D, generator=torch.Generator().manual_seed(4)), torch.randn(B, D, generator=torch.Generator().manual_seed(5)))

    def run(make_opt):
        torch.manual_seed(11)
        m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], num_seqs=S, reference_compat=False).cuda()
        opt = make_opt(m.parameters())
        for _ in range(4):
            opt.zero_grad()
            out = m(x, idx, S, ns, eps=eps)
            loss_function(out[0], out[1], 10.0).backward()
            opt.step()
        return {n: p.detach().clone() for n, p in m.named_parameters()}

    a = run(lambda ps: torch.optim.Adam(ps, lr=1e-3, betas=(0.95, 0.999)))
    b = run(lambda ps: FusedAdam(ps, lr=1e-3, betas=(0.95, 0.999)))
    for n in a:
        close(b[n], a[n], rtol=2e-4, what=n)


def test_edge_cases_single_segment_single_row_and_bad_index(hb):
    """B=1, S=1 (a one-row table: CE must be exactly 0), T=1, and the reference's IndexError for an out-of-range
    sequence index arriving on the host."""
    from fhvae import FHVAE
    from simple_fhvae import SimpleFHVAE

    torch.manual_seed(0)
    m = SimpleFHVAE(4 * 8, [16, 16], [16, 16], 16, 16, [16, 16], num_seqs=1).cuda()
    out = m(torch.randn(1, 4, 8).cuda(), torch.tensor([0]), 1, 5)
    assert out[0].shape == (1,) and abs(out[1].item()) < 1e-6 and torch.isfinite(out[0]).all()
    f = FHVAE(1 * 8, [8], [8], 4, 4, [8], seg_len=1, num_seqs=3).cuda()  # one frame per segment, one layer
    ref = R.FHVAERef(8, [8], [8], 4, 4, [8], seg_len=1)
    ref.load_state_dict({k: v.cpu() for k, v in f.state_dict().items() if k != "mu2_table"})
    x, idx, e2, e1 = torch.randn(2, 1, 8), torch.tensor([2, 0]), torch.randn(2, 4), torch.randn(2, 4)
    got = f(x.cuda(), idx, 3, 7, eps=(e2, e1))
    want = ref(x, idx, 3, 7, mu2_table=f.mu2_table.detach().cpu(), eps_z2=e2, eps_z1=e1)
    for k in range(6):
        close(got[k], want[k], what="out%d" % k)
    with pytest.raises(IndexError):
        f(x.cuda(), torch.tensor([3, 0]), 3, 7)


@pytest.mark.parametrize("dt,tol", [("f32", 2e-3), ("bf16", 2e-2)])
def test_training_trajectory_matches_cpu_oracle(hb, dt, tol):
    """'At matched ELBO': 12 Adam steps of the HIP model (FusedAdam, gradient sinks, persistent table) and of the CPU
    oracle from the same initial weights, data and draws -- the loss and the bound stay together step by step
    (f32: 2e-3 relative; bf16 operands: 2e-2)."""
    from fhvae import FHVAE
    from hip_optim import FusedAdam
    from train_model import loss_function

    T, F, H, D, B, S, steps = 20, 80, 64, 16, 64, 50, 12
    torch.manual_seed(21)
    ref = R.FHVAERef(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T)
    table0 = torch.randn(S, D)
    m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T, num_seqs=S, reference_compat=False, compute_dtype=dt)
    m.load_state_dict(dict(ref.state_dict(), mu2_table=table0.clone()))
    m.cuda()
    opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
    table = table0.clone().requires_grad_(True)
    ropt = torch.optim.Adam(list(ref.parameters()) + [table], lr=1e-3, betas=(0.95, 0.999))
    g = torch.Generator().manual_seed(5)
    got, want = [], []
    for k in range(steps):
        x = torch.randn(B, T, F, generator=g)
        idx = torch.randint(0, S, (B,), generator=g)
        ns = torch.randint(20, 200, (B,), generator=g)
        e2, e1 = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
        loss_ref, lb_ref = R.train_step(ref, ropt, table, x, idx, ns, e2, e1, alpha=10.0, reference_compat=False)
        opt.zero_grad()
        out = m(x.cuda(), idx, S, ns, eps=(e2, e1))
        loss = loss_function(out[0], out[1], 10.0)
        loss.backward()
        opt.step()
        got.append((loss.item(), out[0].mean().item() / T))
        want.append((loss_ref.item(), lb_ref.mean().item() / T))
    for k, ((lg, eg), (lw, ew)) in enumerate(zip(got, want)):
        assert abs(lg - lw) <= tol * abs(lw), ("loss", k, lg, lw)
        assert abs(eg - ew) <= tol * abs(ew), ("elbo nats/frame", k, eg, ew)
    assert want[-1][0] < want[0][0]  # it trains


@pytest.mark.parametrize("dt,tol", [("f32", 5e-3), ("bf16", 3e-2)])
def test_matched_elbo_200_steps_per_term(hb, dt, tol):
    """'At matched ELBO' over a longer horizon: 200 Adam steps of the HIP model and of the CPU oracle from the same initial
    weights, on the same fixed batches (a cycle of 4) and reparameterisation draws.  Compared at steps 50, 100, 150, 200: the
    mean of EVERY term the forward returns -- lower bound, log_qy (-CE), log p(x|z), -KL(z1), -KL(z2), log p(mu2) -- and the loss.
    Training moves them a long way (the bound by tens of nats per segment), so staying together is not an artefact of a short
    horizon.  f32: 5e-3 of each term's magnitude (ulp-level differences in f32 accumulation orders are amplified by 200 Adam
    steps); bf16 operands: 3e-2."""
    from fhvae import FHVAE
    from hip_optim import FusedAdam
    from train_model import loss_function

    T, F, H, D, B, S, steps = 20, 80, 32, 16, 32, 20, 200
    torch.manual_seed(33)
    ref = R.FHVAERef(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T)
    table0 = torch.randn(S, D)
    m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T, num_seqs=S, reference_compat=False, compute_dtype=dt)
    m.load_state_dict(dict(ref.state_dict(), mu2_table=table0.clone()))
    m.cuda()
    opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
    table = table0.clone().requires_grad_(True)
    ropt = torch.optim.Adam(list(ref.parameters()) + [table], lr=1e-3, betas=(0.95, 0.999))
    g = torch.Generator().manual_seed(6)
    batches = [(torch.randn(B, T, F, generator=g), torch.randint(0, S, (B,), generator=g), torch.randint(20, 200, (B,), generator=g))
               for _ in range(4)]
    names = ("lower_bound", "log_qy", "log_px_z", "neg_kld_z1", "neg_kld_z2", "log_pmu2")
    first = None
    for k in range(steps):
        x, idx, ns = batches[k % 4]
        e2, e1 = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
        ropt.zero_grad(set_to_none=True)
        want = ref(x, idx, S, ns, mu2_table=table, eps_z2=e2, eps_z1=e1, reference_compat=False)
        loss_ref = R.loss_function(want[0], want[1], 10.0)
        loss_ref.backward()
        ropt.step()
        opt.zero_grad()
        got = m(x.cuda(), idx, S, ns, eps=(e2, e1))
        loss = loss_function(got[0], got[1], 10.0)
        loss.backward()
        opt.step()
        if first is None:
            first = [w.detach().mean().item() for w in want]
        if (k + 1) % 50 == 0:
            assert abs(loss.item() - loss_ref.item()) <= tol * abs(loss_ref.item()), ("loss", k, loss.item(), loss_ref.item())
            for n, gk, wk in zip(names, got, want):
                gv, wv = gk.detach().mean().item(), wk.detach().mean().item()
                assert abs(gv - wv) <= tol * max(abs(wv), 1.0), (n, k, gv, wv)
    last = [w.detach().mean().item() for w in want]
    assert last[0] > first[0] + 10.0, (first[0], last[0])  # the bound moved by tens of nats per segment over the horizon
