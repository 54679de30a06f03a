"""Checkpoint / resume contract on the GPU (utils.py:63-152, train_model.py:409-415): FusedAdam's state dict carries the
moments and the step count in torch.optim.Adam's layout; a reference-layout checkpoint (fixture written from the imported
reference) loads, runs and resumes; train_model.main's --hip-graph mode trains like the eager loop."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from test_ops_gpu import close, dev, hb  # noqa: F401

OUT = ["lower_bound", "log_qy", "log_px_z", "neg_kld_z1", "neg_kld_z2", "log_pmu2"]


def _step(m, opt, x, idx, S, ns, eps):
    from train_model import loss_function

    opt.zero_grad()
    out = m(x, idx, S, ns, eps=eps)
    loss = loss_function(out[0], out[1], 10.0)
    loss.backward()
    opt.step()
    return loss.item()


def test_fused_adam_state_dict_roundtrip_matches_uninterrupted_run(hb, tmp_path):
    """save after 3 steps -> fresh model + optimizer -> load -> 3 more steps == 6 uninterrupted steps (bit for bit: same
    kernels, same inputs, same state)."""
    import utils as U
    from fhvae import FHVAE
    from hip_optim import FusedAdam

    T, F, H, D, B, S = 20, 80, 32, 16, 32, 40
    g = torch.Generator().manual_seed(3)
    x, idx, ns = torch.randn(B, T, F, generator=g).cuda(), torch.randint(0, S, (B,), generator=g), torch.randint(20, 200, (B,), generator=g)
    eps = [(torch.randn(B, D, generator=g).cuda(), torch.randn(B, D, generator=g).cuda()) for _ in range(6)]

    def build():
        torch.manual_seed(1)
        m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], seg_len=T, num_seqs=S, reference_compat=False).cuda()
        return m, FusedAdam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))

    m_a, opt_a = build()
    la = [_step(m_a, opt_a, x, idx, S, ns, eps[k]) for k in range(6)]
    m_b, opt_b = build()
    lb = [_step(m_b, opt_b, x, idx, S, ns, eps[k]) for k in range(3)]
    sd = opt_b.state_dict()
    assert len(sd["state"]) == len(list(m_b.parameters())) and float(sd["state"][0]["step"]) == 3.0
    assert sd["state"][0]["exp_avg"].shape == next(iter(m_b.parameters())).shape and sd["state"][0]["exp_avg"].abs().sum() > 0
    U.save_checkpoint(m_b, opt_b, None, {}, "run", 0, 0, 0.0, 0.0, str(tmp_path))
    m_c, _, optim_state, start_epoch, _, _ = U.load_checkpoint_file(tmp_path / "fhvae_run_e0.tar", finetune=False)
    m_c.cuda()
    opt_c = FusedAdam(m_c.parameters(), lr=1e-3, betas=(0.95, 0.999))
    opt_c.load_state_dict(optim_state)
    assert int(opt_c.step_dev.item()) == 3 and start_epoch == 1
    lc = [_step(m_c, opt_c, x, idx, S, ns, eps[k]) for k in range(3, 6)]
    assert lb == la[:3] and all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(la[3:], lc)), (la, lb + lc)
    for (n, p), (_, q) in zip(m_a.named_parameters(), m_c.named_parameters()):
        # (gradients are summed with float atomics: the last bits of an update depend on arrival order)
        torch.testing.assert_close(p, q, rtol=1e-5, atol=2e-6, msg=lambda s: n + ": " + s)
    # and torch.optim.Adam's own state dict over the same parameters loads too (same layout)
    t_opt = torch.optim.Adam(m_a.parameters(), lr=1e-3, betas=(0.95, 0.999))
    t_opt.load_state_dict(opt_a.state_dict())
    assert float(t_opt.state[next(iter(m_a.parameters()))]["step"]) == 6.0


def test_reference_checkpoint_runs_and_resumes(hb, golden_dir):
    """The reference-layout fixture: loaded model reproduces the reference's forward outputs on the GPU; its Adam state
    (torch.optim.Adam, 16 of 24 tensors with moments) loads into FusedAdam and the next step follows the CPU oracle
    resumed from the same state with torch.optim.Adam."""
    import utils as U
    from hip_optim import FusedAdam
    from train_model import loss_function

    f = os.path.join(golden_dir, "ref_checkpoint_simple_tiny.tar")
    g = dict(np.load(os.path.join(golden_dir, "simple_tiny_f32.npz")))
    o = dict(np.load(os.path.join(golden_dir, "ref_checkpoint_simple_tiny_outputs.npz")))
    m, _, optim_state, _, _, _ = U.load_checkpoint_file(f, finetune=False, input_size=32)
    m.cuda()
    x, idx, ns = torch.from_numpy(g["x"]), torch.from_numpy(g["idx"]), torch.from_numpy(g["nsegs"])
    table, e2, e1 = torch.from_numpy(o["table"]), torch.from_numpy(o["eps_z2"]), torch.from_numpy(o["eps_z1"])
    with torch.no_grad():
        got = m(dev(x), idx, 12, ns, mu2_table=dev(table), eps=(e2, e1))
    for k, n in enumerate(OUT):
        close(got[k], torch.from_numpy(o["out_" + n]), what=n)
    # resume: one more step, GPU (FusedAdam) vs CPU oracle (torch.optim.Adam), both from the checkpoint's optimizer state
    ref = R.SimpleFHVAERef(32, [16, 16], [16, 16], 16, 16, [16, 16])
    ref.load_state_dict(torch.load(f, weights_only=False)["state_dict"])
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.95, 0.999))
    import copy

    ropt.load_state_dict(copy.deepcopy(optim_state))  # (torch's Adam keeps and increments the loaded `step` tensors in place)
    ropt.zero_grad()
    want = ref(x, idx, 12, ns, mu2_table=table, eps_z2=e2, eps_z1=e1, reference_compat=True)
    R.loss_function(want[0], want[1], 10.0).backward()
    ropt.step()
    opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
    opt.load_state_dict(optim_state)
    assert int(opt.step_dev.item()) == 1
    opt.zero_grad()
    got = m(dev(x), idx, 12, ns, mu2_table=dev(table), eps=(e2, e1))
    loss_function(got[0], got[1], 10.0).backward()
    opt.step()
    rp = dict(ref.named_parameters())
    for n, p in m.named_parameters():
        if n.startswith(("pre_decoder", "dec_gauss_layer")):
            assert torch.equal(p.detach().cpu(), rp[n].detach()), n  # no gradient, no moments: untouched by both
        else:
            torch.testing.assert_close(p.detach().cpu(), rp[n].detach(), rtol=1e-5, atol=2e-6, msg=lambda s: n + ": " + s)


def _epoch_losses(text):
    return [float(v) for v in re.findall(r"Train set average loss: (-?[0-9.]+|nan|inf)", text)]


def test_train_model_hip_graph_matches_eager(hb, capsys):
    """train_model.main(--hip-graph): the captured step is replayed for every batch, its outputs are read only after a
    replay (they were read straight after the capture = uninitialised memory), warm-up does not train.  The
    reparameterisation draws differ between the two modes (graph-safe generator offsets), so the epoch losses agree to
    sampling noise (2 %), not bitwise."""
    import train_model as TM

    base = ["--z1-hus", "64", "64", "--z2-hus", "64", "64", "--x-hus", "64", "64", "--z1-dim", "16", "--z2-dim", "16",
            "--epochs", "2", "--train-segments", "256", "--dev-segments", "64", "--training-batch-size", "64",
            "--num-seqs", "30", "--seed", "5"]
    assert TM.main(base) == 0
    eager = capsys.readouterr().out
    assert TM.main(base + ["--hip-graph"]) == 0
    graphed = capsys.readouterr().out
    le, lg = _epoch_losses(eager), _epoch_losses(graphed)
    assert len(le) == 2 and len(lg) == 2 and all(np.isfinite(le + lg)), (le, lg)
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-2 * abs(a), (le, lg)
    assert "Training diverged" not in graphed
    # bf16 + graph + persistent recurrence kernels (needs the whole GPU): also finite, status words clean
    assert TM.main(["--z1-hus", "256", "256", "--z2-hus", "256", "256", "--x-hus", "256", "256", "--z1-dim", "32", "--z2-dim", "32",
                    "--epochs", "1", "--train-segments", "512", "--dev-segments", "64", "--training-batch-size", "256",
                    "--num-seqs", "30", "--compute-dtype", "bf16", "--hip-graph"]) == 0
    assert hb.lstm_sync_status() == 0 and not hb.diverged()


def test_divergence_flag_is_sticky_and_sync_free(hb):
    """fhvae_loss_fwd records a NaN lower bound in the device word; nothing is read back until asked."""
    from train_model import loss_function

    hb.reset_device_words()
    lb = torch.randn(64, device="cuda")
    qy = torch.zeros((), device="cuda")
    loss_function(lb, qy, 10.0)
    assert not hb.diverged()
    lb[5] = float("nan")
    loss_function(lb, qy, 10.0)
    loss_function(torch.randn(64, device="cuda"), qy, 10.0)  # a later healthy batch does not clear it
    assert hb.diverged()
    hb.reset_device_words()
    assert not hb.diverged()
