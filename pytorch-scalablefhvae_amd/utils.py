"""utils.py -- the parts of the reference's utils.py that touch the hot path's state (utils.py:14-152):
checkpoint layout, args pickle, and the closed-form mu2 estimate (run naming, utils.py:20-42, is CLI plumbing: not built).  AudioUtils (librosa wrappers,
utils.py:155-300) is offline feature extraction and out of scope.
"""
from __future__ import annotations

import pickle
import shutil
from pathlib import Path

import numpy as np
import torch

from fhvae import FHVAE
from simple_fhvae import SimpleFHVAE


def check_best(val_lower_bound, best_val_lb) -> bool:
    """utils.py:14-17."""
    return bool(torch.mean(val_lower_bound) > best_val_lb)


def estimate_mu2_dict(model, loader, num_seqs):
    """Estimate mu2 for sequences (utils.py:45-60): mu2[y] = sum z2_mu / (n_y + exp(pz2_logvar)/exp(pmu2_logvar)).
    `loader` yields (idxs, features, nsegs) like the reference's DataLoader (or ResidentSegmentPool.epoch).
    The per-sequence sums run on the device (fhvae_mu2_accumulate / _finalize); returns {y: (D,) tensor} for the
    sequences that occurred, like the reference."""
    import hip_binding as hb

    model.eval()
    est = None
    with torch.no_grad():
        for idxs, features, nsegs in loader:
            _, z2_mu = model.encode(features)
            if est is None:
                est = hb.Mu2Estimator(num_seqs, z2_mu.shape[1], z2_mu.device)
            est.add(z2_mu, torch.as_tensor(idxs).to(device=z2_mu.device, dtype=torch.int64))
    r = float(np.exp(model.pz2[1]) / np.exp(model.pmu2[1]))  # utils.py:58
    mu2, count = est.result(r)
    seen = torch.nonzero(count > 0).flatten().tolist()
    return {y: mu2[y] for y in seen}


def save_args(exp_dir, args):
    with open(f"{exp_dir}/args.pkl", "wb") as f:
        pickle.dump(args, f)


def load_args(exp_dir):
    with open(f"{exp_dir}/args.pkl", "rb") as f:
        return pickle.load(f)


def save_checkpoint(model, optimizer, summary_list, values_dict, run_info: str, epoch: int, best_epoch: int,
                    val_lower_bound: float, best_val_lb: float, checkpoint_dir: str, input_size=None) -> None:
    """Same dict layout and file names as utils.py:116-152.  `model_params` additionally carries the input size as
    its first element (the reference stores 5 values but both constructors take 6, utils.py:75,135-141); the mu2
    table travels inside `state_dict` (key `mu2_table`)."""
    if input_size is None:
        input_size = getattr(model, "seg_len", 1) * getattr(model, "n_feat", 0) or model.dec_gauss_layer.mulayer.out_features
    checkpoint = {
        "best_val_lb": best_val_lb,
        "best_epoch": best_epoch,
        "epoch": epoch,
        "model_type": model.model,
        "model_params": (input_size, model.z1_hus, model.z2_hus, model.z1_dim, model.z2_dim, model.x_hus),
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        "summary_vals": summary_list,
        "values": values_dict,
        # one key beyond the reference's layout: keyword-only constructor arguments of this build
        "model_kwargs": {k: getattr(model, k) for k in ("seg_len", "compute_dtype", "reference_compat") if hasattr(model, k)},
    }
    f_str = f"{model.model}_{run_info}_e{epoch}"
    f_path = Path(checkpoint_dir) / f"{f_str}.tar"
    torch.save(checkpoint, f_path)
    if best_epoch == epoch:
        shutil.copyfile(f_path, Path(checkpoint_dir) / f"best_model_{f_str}.tar")


def load_checkpoint_file(checkpoint_file, finetune, input_size=None):
    """utils.py:63-102.  Accepts the reference's 5-value `model_params` (then `input_size` must be given) and this
    build's 6-value form; a reference checkpoint has no `mu2_table` (the reference never kept one)."""
    optim_state = start_epoch = best_val_lb = summary_list = values = None
    checkpoint = torch.load(checkpoint_file, map_location="cpu", weights_only=False)
    model_type = checkpoint["model_type"]
    params = tuple(checkpoint["model_params"])
    if len(params) == 5:
        if input_size is None:
            raise ValueError("reference-style checkpoint (5 model_params): pass input_size=")
        params = (input_size,) + params
    sd = checkpoint["state_dict"]
    kw = dict(checkpoint.get("model_kwargs", {}))
    if model_type != "fhvae":
        kw.pop("seg_len", None), kw.pop("compute_dtype", None)
    if "mu2_table" in sd:
        kw["num_seqs"] = sd["mu2_table"].shape[0]
    if model_type == "fhvae":
        model = FHVAE(*params, **kw)
    elif model_type == "simple_fhvae":
        model = SimpleFHVAE(*params, **kw)
    else:
        raise ValueError(f"NON-STANDARD MODEL TYPE {model_type}")
    model.load_state_dict(sd, strict="mu2_table" in sd)
    if not finetune:
        optim_state = checkpoint["optimizer"]
        start_epoch = checkpoint["epoch"] + 1  # saved at the end of an epoch (the reference adds 1 twice, utils.py:89,92)
        best_val_lb = checkpoint["best_val_lb"]
        summary_list = checkpoint["summary_vals"]
        values = checkpoint["values"]
    return model, values, optim_state, start_epoch, best_val_lb, summary_list
