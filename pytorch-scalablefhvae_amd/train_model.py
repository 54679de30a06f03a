"""train_model.py -- the reference's training loop shape (train_model.py:243-261, :438-541) on the HIP
hot path.  Importing this module has no side effects (the reference parses argv at import time,
train_model.py:25-238, which makes `loss_function` un-importable; the CLI lives in `main()` here).

Kept from the reference: flag names and defaults of the flags the hot path reads (SURVEY section 2
row 5), `loss_function`, `check_terminate`, `check_best`, the per-batch order zero_grad -> forward ->
loss -> backward -> step, the NaN abort with exit code 2 (train_model.py:464-466), Adam with
betas (0.95, 0.999).  Not reproduced: the reference's defects listed in SURVEY 3.1 (fp64 cast,
wrong validation loss, double division, ...).  Data: synthetic (B,T,F) segments by default; real
feature scp files are SURVEY 8f "next" #1.
"""
from __future__ import annotations

import argparse
import sys
import time
from typing import Optional

import numpy as np
import torch


# alpha/discriminative weight of 10 was found to produce best results (train_model.py:240)
def loss_function(lower_bound, log_qy, alpha=10.0):
    """Discriminative segment variational lower bound: -mean(lower_bound + alpha*log_qy)
    (train_model.py:243-251)."""
    return -1 * torch.mean(lower_bound + alpha * log_qy)


def check_terminate(epoch, best_epoch, patience, epochs):
    """train_model.py:254-261."""
    if (epoch - 1) - best_epoch > patience:
        return True
    if epoch > epochs:
        return True
    return False


def check_best(val_lower_bound, best_val_lb) -> bool:
    """utils.py:14-17."""
    return bool(torch.mean(val_lower_bound) > best_val_lb)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="ScalableFHVAE training on MI355X (HIP hot path)")
    p.add_argument("--model-type", default="fhvae", choices=["fhvae", "simple_fhvae"])  # train_model.py:140-144
    p.add_argument("--alpha-dis", type=float, default=10.0)                              # :48-53
    p.add_argument("--z1-hus", nargs=2, default=[128, 128])                               # :145-150 (strings from CLI)
    p.add_argument("--z2-hus", nargs=2, default=[128, 128])
    p.add_argument("--z1-dim", type=int, default=16)
    p.add_argument("--z2-dim", type=int, default=16)
    p.add_argument("--x-hus", nargs=2, default=[128, 128])
    p.add_argument("--seg-len", type=int, default=20)                                     # :120-123
    p.add_argument("--mels", type=int, default=80)
    p.add_argument("--training-batch-size", type=int, default=256)                        # :134-137
    p.add_argument("--dev-batch-size", type=int, default=2048)
    p.add_argument("--learning-rate", type=float, default=1e-3)
    p.add_argument("--beta-one", type=float, default=0.95)
    p.add_argument("--beta-two", type=float, default=0.999)
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--patience", type=int, default=10)
    p.add_argument("--device", default="gpu")
    # synthetic-data controls (no reference counterpart: the reference reads scp files)
    p.add_argument("--num-seqs", type=int, default=100)
    p.add_argument("--train-segments", type=int, default=1000)
    p.add_argument("--dev-segments", type=int, default=250)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--paper-objective", action="store_true",
                   help="train the intended objective (decoder attached, log_qy=-CE) instead of the reference's literal one")
    return p


def synthetic_split(n, T, F, S, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, T, F, generator=g)
    idx = torch.randint(0, S, (n,), generator=g)
    nsegs = torch.randint(20, 200, (n,), generator=g)
    return x, idx, nsegs


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    if args.device != "gpu" or not torch.cuda.is_available():
        print("this training path runs on a MI355X only (no CPU fallback)", file=sys.stderr)
        return 1
    device = torch.device("cuda:0")
    from fhvae import FHVAE
    from simple_fhvae import SimpleFHVAE

    torch.manual_seed(args.seed)
    T, F, S = args.seg_len, args.mels, args.num_seqs
    input_size = T * F  # np.prod(example_data.shape), train_model.py:396-398
    kw = dict(num_seqs=S, reference_compat=not args.paper_objective)
    if args.model_type == "fhvae":
        model = FHVAE(input_size, args.z1_hus, args.z2_hus, args.z1_dim, args.z2_dim, args.x_hus, seg_len=T, **kw)
    else:
        model = SimpleFHVAE(input_size, args.z1_hus, args.z2_hus, args.z1_dim, args.z2_dim, args.x_hus, **kw)
    model.to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.learning_rate, betas=(args.beta_one, args.beta_two))

    xtr, itr, ntr = synthetic_split(args.train_segments, T, F, S, args.seed + 1)
    xdv, idv, ndv = synthetic_split(args.dev_segments, T, F, S, args.seed + 2)
    xtr, xdv = xtr.to(device), xdv.to(device)
    best_epoch, best_val_lb = 0, -np.inf
    for epoch in range(args.epochs):
        model.train()
        t0 = time.time()
        train_loss = torch.zeros((), device=device)
        perm = torch.randperm(xtr.shape[0])
        nb = 0
        for s in range(0, xtr.shape[0], args.training_batch_size):
            sel = perm[s:s + args.training_batch_size]
            optimizer.zero_grad()
            lower_bound, discrim_loss, log_px_z, neg_kld_z1, neg_kld_z2, log_pmu2 = model(xtr[sel], itr[sel], S, ntr[sel])
            loss = loss_function(lower_bound, discrim_loss, args.alpha_dis)
            loss.backward()
            optimizer.step()
            train_loss += loss.detach()
            nb += 1
            if torch.isnan(lower_bound).any():
                print("Training diverged")
                return 2  # sys.exit(2), train_model.py:464-466
        dt = time.time() - t0
        print(f"====> Train set average loss: {train_loss.item() / nb:.4f}  ({xtr.shape[0] / dt:.0f} segments/s)")
        model.eval()
        lbs = []
        with torch.no_grad():
            for s in range(0, xdv.shape[0], args.dev_batch_size):
                out = model(xdv[s:s + args.dev_batch_size], idv[s:s + args.dev_batch_size], S, ndv[s:s + args.dev_batch_size])
                lbs.append(out[0])
        val_lower_bound = torch.cat(lbs)
        print(f"====> Validation set lower bound: {val_lower_bound.mean().item():.4f} "
              f"({val_lower_bound.mean().item() / T:.4f} nats/frame)")
        if check_best(val_lower_bound, best_val_lb):
            best_epoch, best_val_lb = epoch, val_lower_bound.mean().item()
        if check_terminate(epoch, best_epoch, args.patience, args.epochs):
            print("Training terminated!")
            break
    print("Training complete!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
