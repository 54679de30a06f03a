#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (needs ``/root/reference``; the GPU box never sees it):

    python tests/golden/make_golden.py

What it does: imports the reference's ``simple_fhvae.SimpleFHVAE`` (simple_fhvae.py:8-124) under a
private module name, calls its ``forward`` on seeded inputs, and stores inputs + the random draws
the reference made + outputs (+ gradients) as ``.npz``.  The random draws are captured *from the
reference call itself* (a wrapper around ``mu2_lookup`` and ``torch.randn_like`` records what
was drawn; nothing is replaced), so the fixtures are data only -- no reference source text.

Fixtures:
  simple_tiny_f32.npz / simple_tiny_f64.npz : T=4,F=8,hus=16/16,D=16,B=8,S=12; state_dict, inputs,
        draws, the 6 forward outputs, loss_function(alpha=10) and all gradients (+ table.grad).
  simple_refshape_f32.npz : T=20,F=80, hus 128/128, D=16, B=64, S=100; weights/x from the closed-form
        ``oracle.ref_cpu.det_tensor`` (regenerable), draws + 6 outputs + loss stored.
  ref_checkpoint_simple_tiny.tar : a checkpoint in the reference's EXACT dict layout (utils.py:131-146: 5-value
        `model_params`, plain `state_dict()`, `optimizer.state_dict()` of torch.optim.Adam(lr, betas=(0.95, 0.999)),
        train_model.py:409-411) written from the imported reference model after one reference training step on the
        tiny fixture's inputs; plus the 6 forward outputs of the UPDATED model on the same inputs/draws.  (The reference's
        own `utils.save_checkpoint` cannot be imported -- utils.py:1,4 need librosa/nptyping -- so the dict literal
        follows utils.py:131-146 key by key; the values all come from reference objects.)
  disc_{8x12,256x4600}.npz : the discriminative block alone (simple_fhvae.py:119-122): q=z2_mu, table,
        idx -> log_qy, d log_qy/dq, d log_qy/dtable, taken from inside a reference forward.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.ref_cpu import det_index, det_tensor, fill_state_dict_det  # noqa: E402

REF = "/root/reference/simple_fhvae.py"
spec = importlib.util.spec_from_file_location("_reference_simple_fhvae", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def ref_loss_function(lower_bound, log_qy, alpha=10.0):
    # train_model.py:243-251 (train_model cannot be imported: argparse + missing deps at import)
    return -1 * torch.mean(lower_bound + alpha * log_qy)


class Recorder:
    """Records the reference's own random draws and the z2_mu it computed."""

    def __init__(self, model):
        self.model = model
        self.draws = []
        self.table = None
        self.z2_mu = None
        orig_lookup = model.mu2_lookup

        def lookup(*a, **k):
            table, mu2 = orig_lookup(*a, **k)
            self.table = table
            return table, mu2

        model.mu2_lookup = lookup
        self._orig_randn_like = torch.randn_like

        def randn_like(t, *a, **k):
            e = self._orig_randn_like(t, *a, **k)
            self.draws.append(e)
            return e

        self._patched = randn_like

        def hook(_m, _i, out):
            self.z2_mu = out[0]
            if out[0].requires_grad:
                out[0].retain_grad()

        model.z2_gauss_layer.register_forward_hook(hook)

    def __enter__(self):
        torch.randn_like = self._patched
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig_randn_like


def run(model, x, idx, S, nsegs, seed):
    torch.manual_seed(seed)
    with Recorder(model) as rec:
        out = model(x, idx, S, nsegs)
    # draw order: table, eps_z2, eps_z1, eps_dec (unused x_sample)
    return out, rec


def np_(t):
    return t.detach().cpu().numpy()


def tiny(dtype, name):
    T, F, D, B, S = 4, 8, 16, 8, 12
    torch.manual_seed(7)
    model = ref.SimpleFHVAE(T * F, [16, 16], [16, 16], D, D, [16, 16])
    if dtype == torch.float64:
        model.double()
    x = torch.randn(B, T, F, dtype=dtype)
    idx = torch.randint(0, S, (B,))
    idx[1] = idx[0]  # force a duplicate index (gather collision in table.grad)
    nsegs = torch.randint(3, 40, (B,))
    out, rec = run(model, x, idx, S, nsegs, seed=11)
    loss = ref_loss_function(out[0], out[1], 10.0)
    loss.backward()
    d = {"x": np_(x), "idx": np_(idx), "nsegs": np_(nsegs), "table": np_(rec.table),
         "eps_z2": np_(rec.draws[0]), "eps_z1": np_(rec.draws[1]), "loss": np_(loss), "alpha": np.float64(10.0)}
    for k, name_o in enumerate(["lower_bound", "log_qy", "log_px_z", "neg_kld_z1", "neg_kld_z2", "log_pmu2"]):
        d["out_" + name_o] = np_(out[k])
    for n, p in model.state_dict().items():
        d["sd_" + n] = np_(p)
    for n, p in model.named_parameters():
        d["grad_" + n] = np_(p.grad) if p.grad is not None else np.zeros(0)
    d["grad_table"] = np_(rec.table.grad)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "loss", float(loss), "no-grad params:", [n for n, p in model.named_parameters() if p.grad is None])


def ref_checkpoint():
    T, F, D, B, S = 4, 8, 16, 8, 12
    g = dict(np.load(os.path.join(HERE, "simple_tiny_f32.npz")))
    torch.manual_seed(7)
    model = ref.SimpleFHVAE(T * F, [16, 16], [16, 16], D, D, [16, 16])
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")})
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.95, 0.999))  # train_model.py:409-411
    x, idx, nsegs = torch.from_numpy(g["x"]), torch.from_numpy(g["idx"]), torch.from_numpy(g["nsegs"])
    optimizer.zero_grad()
    out, rec = run(model, x, idx, S, nsegs, seed=11)        # same seed as tiny(): same table and draws
    loss = ref_loss_function(out[0], out[1], 10.0)
    loss.backward()
    optimizer.step()                                        # train_model.py:446-454
    with torch.no_grad():
        out2, rec2 = run(model, x, idx, S, nsegs, seed=11)  # the updated model on the same inputs/draws
    checkpoint = {                                          # utils.py:131-146, key by key
        "best_val_lb": float(torch.mean(out2[0])),
        "best_epoch": 3,
        "epoch": 3,
        "model_type": model.model,
        "model_params": (model.z1_hus, model.z2_hus, model.z1_dim, model.z2_dim, model.x_hus),
        "optimizer": optimizer.state_dict(),
        "state_dict": model.state_dict(),
        "summary_vals": None,
        "values": {"val_lower_bound": float(torch.mean(out2[0]))},
    }
    torch.save(checkpoint, os.path.join(HERE, "ref_checkpoint_simple_tiny.tar"))
    d = {"table": np_(rec2.table), "eps_z2": np_(rec2.draws[0]), "eps_z1": np_(rec2.draws[1])}
    for k, name_o in enumerate(["lower_bound", "log_qy", "log_px_z", "neg_kld_z1", "neg_kld_z2", "log_pmu2"]):
        d["out_" + name_o] = np_(out2[k])
    np.savez_compressed(os.path.join(HERE, "ref_checkpoint_simple_tiny_outputs.npz"), **d)
    print("ref_checkpoint_simple_tiny.tar: loss before the step", float(loss), "params with Adam state:",
          len(checkpoint["optimizer"]["state"]))


def refshape():
    T, F, D, B, S = 20, 80, 16, 64, 100
    model = ref.SimpleFHVAE(T * F)  # defaults 128/128, 16/16  (simple_fhvae.py:9-17)
    fill_state_dict_det(model, seed=1.0)
    x = det_tensor((B, T, F), seed=3.0)
    idx = det_index(B, S, seed=5)
    nsegs = det_index(B, 180, seed=9) + 20
    with torch.no_grad():
        out, rec = run(model, x, idx, S, nsegs, seed=21)
        loss = ref_loss_function(out[0], out[1], 10.0)
    d = {"table": np_(rec.table), "eps_z2": np_(rec.draws[0]), "eps_z1": np_(rec.draws[1]), "loss": np_(loss),
         "meta_TFDBS": np.array([T, F, D, B, S])}
    for k, name_o in enumerate(["lower_bound", "log_qy", "log_px_z", "neg_kld_z1", "neg_kld_z2", "log_pmu2"]):
        d["out_" + name_o] = np_(out[k])
    np.savez_compressed(os.path.join(HERE, "simple_refshape_f32.npz"), **d)
    print("simple_refshape_f32.npz loss", float(loss))


def disc(B, S, name):
    T, F, D = 4, 8, 16
    torch.manual_seed(100 + B)
    model = ref.SimpleFHVAE(T * F, [16, 16], [16, 16], D, D, [16, 16])
    with torch.no_grad():  # widen z2_mu so the logits span a realistic range (hundreds of nats)
        model.z2_gauss_layer.mulayer.weight.mul_(6.0)
    x = torch.randn(B, T, F)
    idx = torch.randint(0, S, (B,))
    nsegs = torch.randint(3, 40, (B,))
    out, rec = run(model, x, idx, S, nsegs, seed=300 + B)
    out[1].backward()  # log_qy alone: depends on (z2_mu, table, idx) only (simple_fhvae.py:119-122)
    np.savez_compressed(
        os.path.join(HERE, name), q=np_(rec.z2_mu), table=np_(rec.table), idx=np_(idx), log_qy=np_(out[1]),
        dq=np_(rec.z2_mu.grad), dtable=np_(rec.table.grad))
    print(name, "log_qy", float(out[1]))


if __name__ == "__main__":
    if "--checkpoint-only" in sys.argv:
        ref_checkpoint()
        sys.exit(0)
    tiny(torch.float32, "simple_tiny_f32.npz")
    tiny(torch.float64, "simple_tiny_f64.npz")
    refshape()
    disc(8, 12, "disc_8x12.npz")
    disc(256, 4600, "disc_256x4600.npz")
    ref_checkpoint()
