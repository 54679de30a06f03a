"""The CPU oracle (oracle/ref_cpu.py) against the golden vectors taken from the reference
(tests/golden/make_golden.py).  Pins the restatement: forward 6-tuple, loss and gradients."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

OUT = ["lower_bound", "log_qy", "log_px_z", "neg_kld_z1", "neg_kld_z2", "log_pmu2"]


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


@pytest.mark.parametrize("name,dtype", [("simple_tiny_f32.npz", torch.float32), ("simple_tiny_f64.npz", torch.float64)])
def test_simple_tiny_forward_loss_grads(golden_dir, name, dtype):
    g = _load(golden_dir, name)
    m = R.SimpleFHVAERef(4 * 8, [16, 16], [16, 16], 16, 16, [16, 16]).to(dtype)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")})
    table = torch.from_numpy(g["table"]).requires_grad_(True)  # stays fp32 in the fp64 run too
    out = m(torch.from_numpy(g["x"]), torch.from_numpy(g["idx"]), 12, torch.from_numpy(g["nsegs"]),
            mu2_table=table, eps_z2=torch.from_numpy(g["eps_z2"]), eps_z1=torch.from_numpy(g["eps_z1"]))
    for k, n in enumerate(OUT):
        # restatement is op-for-op: bit-exact on the same torch build
        np.testing.assert_array_equal(out[k].detach().numpy(), g["out_" + n], err_msg=n)
    loss = R.loss_function(out[0], out[1], float(g["alpha"]))
    np.testing.assert_array_equal(loss.detach().numpy(), g["loss"])
    loss.backward()
    for n, p in m.named_parameters():
        ref = g["grad_" + n]
        if ref.size == 0:  # reference: decoder gets no gradient (detach at simple_fhvae.py:113-115)
            assert p.grad is None, n
        else:
            np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-6, atol=1e-7, err_msg=n)
    np.testing.assert_allclose(table.grad.numpy(), g["grad_table"], rtol=1e-6, atol=1e-7)


def test_simple_refshape_forward(golden_dir):
    g = _load(golden_dir, "simple_refshape_f32.npz")
    T, F, D, B, S = [int(v) for v in g["meta_TFDBS"]]
    m = R.SimpleFHVAERef(T * F)
    R.fill_state_dict_det(m, seed=1.0)
    x = R.det_tensor((B, T, F), seed=3.0)
    idx = R.det_index(B, S, seed=5)
    nsegs = R.det_index(B, 180, seed=9) + 20
    with torch.no_grad():
        out = m(x, idx, S, nsegs, mu2_table=torch.from_numpy(g["table"]), eps_z2=torch.from_numpy(g["eps_z2"]),
                eps_z1=torch.from_numpy(g["eps_z1"]))
    for k, n in enumerate(OUT):
        np.testing.assert_array_equal(out[k].numpy(), g["out_" + n], err_msg=n)
    np.testing.assert_array_equal(R.loss_function(out[0], out[1]).numpy(), g["loss"])


@pytest.mark.parametrize("name", ["disc_8x12.npz", "disc_256x4600.npz"])
def test_disc_block(golden_dir, name):
    g = _load(golden_dir, name)
    q = torch.from_numpy(g["q"]).requires_grad_(True)
    t = torch.from_numpy(g["table"]).requires_grad_(True)
    idx = torch.from_numpy(g["idx"])
    v = R.disc_loss(q, t, idx)
    np.testing.assert_array_equal(v.detach().numpy(), g["log_qy"])
    v.backward()
    np.testing.assert_allclose(q.grad.numpy(), g["dq"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(t.grad.numpy(), g["dtable"], rtol=1e-6, atol=1e-8)
    # the chunked (no (B,S,D) temporary) form agrees to fp32 rounding
    with torch.no_grad():
        vc = R.disc_loss_chunked(q, t, idx, chunk=7 if t.shape[0] < 100 else 1000)
    np.testing.assert_allclose(vc.numpy(), g["log_qy"], rtol=2e-6)


def test_modes_and_helpers():
    torch.manual_seed(0)
    m = R.FHVAERef(4 * 6, [8, 8], [8, 8], 4, 4, [8, 8], seg_len=4)
    B, S = 5, 9
    x = torch.randn(B, 4, 6)
    idx = torch.tensor([0, 3, 3, 8, 1])
    table = torch.randn(S, 4, requires_grad=True)
    kw = dict(mu2_table=table, eps_z2=torch.randn(B, 4), eps_z1=torch.randn(B, 4))
    a = m(x, idx, S, torch.tensor([3, 4, 5, 6, 7]), reference_compat=True, **kw)
    b = m(x, idx, S, torch.tensor([3, 4, 5, 6, 7]), reference_compat=False, **kw)
    for k in (0, 2, 3, 4, 5):
        torch.testing.assert_close(a[k], b[k], rtol=1e-6, atol=0)  # numpy-exp vs torch.exp: 1 ulp
    torch.testing.assert_close(a[1], -b[1], rtol=0, atol=0)
    assert not a[2].requires_grad and b[2].requires_grad  # log_px_z detached only in reference mode
    assert R.check_terminate(12, 0, 10, 100) and not R.check_terminate(5, 0, 10, 100) and R.check_terminate(101, 100, 10, 100)
    assert R.check_best(torch.tensor([1.0, 3.0]), 1.5) and not R.check_best(torch.tensor([1.0]), 1.5)
    mu2, n = R.estimate_mu2(torch.ones(4, 2), torch.tensor([1, 1, 2, 1]), 4)
    torch.testing.assert_close(n, torch.tensor([0.0, 3.0, 1.0, 0.0]))
    torch.testing.assert_close(mu2[1], torch.full((2,), 3.0 / 3.25))
