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
    if (isinstance(lower_bound, torch.Tensor) and isinstance(log_qy, torch.Tensor) and lower_bound.is_cuda and log_qy.is_cuda
            and lower_bound.dim() == 1 and log_qy.dim() == 0 and lower_bound.dtype == torch.float32 and log_qy.dtype == torch.float32):
        import hip_binding as hb  # the model's own outputs on the GPU: the same expression in one launch each way

        return hb.fused_loss(lower_bound, log_qy, alpha)
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
    # real features in the reference's on-disk format (feats.scp / len.scp of .npy files, prepare_numpy_data.py:115-119)
    p.add_argument("--train-feat-scp", default=None)
    p.add_argument("--train-len-scp", default=None)
    p.add_argument("--dev-feat-scp", default=None)
    p.add_argument("--dev-len-scp", default=None)
    p.add_argument("--min-len", type=int, default=None)            # train_model.py:106-112 (defaults to seg_len, :267-268)
    p.add_argument("--mvn-path", default=None)                     # :113-119
    p.add_argument("--seg-shift", type=int, default=8)             # :124-126
    p.add_argument("--rand-seg", action="store_true")
    p.add_argument("--exp-dir", default=None, help="write the reference-layout checkpoint there after every epoch")
    p.add_argument("--hierarchical", dest="sample_hierarchical", action="store_true",   # train_model.py:203-214
                   help="re-estimate the mu2 table in closed form from the encoder before training (utils.py:45-60)")
    p.add_argument("--compute-dtype", default="f32", choices=["f32", "bf16"])
    p.add_argument("--hip-graph", action="store_true",
                   help="capture one training step (zero_grad, forward, loss, backward, Adam) into a hipGraph and replay it for "
                        "every full-size batch (static input buffers; a smaller last batch runs eagerly)")
    p.add_argument("--reference-objective", action="store_true",
                   help="train the reference's LITERAL objective: decoder outputs and mu2 detached inside the bound "
                        "(simple_fhvae.py:107,114) and log_qy = +CE (:122).  With a persistent learnable mu2 table that "
                        "objective pushes CE up without bound and gives the decoder no gradient; the default is the intended "
                        "objective (decoder trained, log_qy = -CE), which is also what bench.py and the README figures use")
    p.add_argument("--paper-objective", action="store_true", help="(accepted for compatibility: this is the default now)")
    p.add_argument("--continue-from", default=None,                 # train_model.py:192-197
                   help="checkpoint file (utils.save_checkpoint layout) to resume from: model, mu2 table, Adam moments and step")
    p.add_argument("--check-interval", type=int, default=100,
                   help="batches between reads of the device-side divergence / recurrence-status words (each read is a host "
                        "sync; they are always read at the end of an epoch and before a checkpoint)")
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

    import hip_binding as hb

    # the sticky status words (divergence / a recurrence launch that gave up) are per process: a run starts clean.  After a run
    # that returned 2 or 3 the model it trained is invalid (NaN updates may have been applied until the check interval caught them)
    hb.reset_device_words(device)

    torch.manual_seed(args.seed)
    T, F = args.seg_len, args.mels
    real = args.train_feat_scp is not None
    if real:
        from datasets import NumpyDataset, ResidentSegmentPool

        min_len = args.min_len if args.min_len is not None else T  # train_model.py:267-268
        tr_ds = NumpyDataset(args.train_feat_scp, args.train_len_scp, min_len, args.mvn_path, T, args.seg_shift, args.rand_seg)
        dv_ds = NumpyDataset(args.dev_feat_scp or args.train_feat_scp, args.dev_len_scp or args.train_len_scp, min_len,
                             args.mvn_path, T, args.seg_shift, False)
        tr_pool, dv_pool = ResidentSegmentPool(tr_ds, device), ResidentSegmentPool(dv_ds, device)
        F = tr_pool.pool.shape[1]
        S = len(tr_ds)  # len(train_loader.dataset), train_model.py:448
    else:
        S = args.num_seqs
    input_size = T * F  # np.prod(example_data.shape), train_model.py:396-398
    kw = dict(num_seqs=S, reference_compat=bool(args.reference_objective))
    if args.reference_objective:
        print("WARNING: --reference-objective trains the reference's literal loss (+CE, detached decoder); "
              "throughput/ELBO figures of this build use the default objective", file=sys.stderr)
    if args.model_type == "fhvae":
        model = FHVAE(input_size, args.z1_hus, args.z2_hus, args.z1_dim, args.z2_dim, args.x_hus, seg_len=T,
                      compute_dtype=args.compute_dtype, **kw)
    else:
        model = SimpleFHVAE(input_size, args.z1_hus, args.z2_hus, args.z1_dim, args.z2_dim, args.x_hus, **kw)
    model.to(device)
    from hip_optim import FusedAdam

    optimizer = FusedAdam(model.parameters(), lr=args.learning_rate, betas=(args.beta_one, args.beta_two))
    import hip_binding as hb

    start_epoch = 0
    best_epoch, best_val_lb = 0, -np.inf
    if args.continue_from:
        # resume (train_model.py:303-322 -> utils.load_checkpoint_file): weights + table into the live model, Adam moments and
        # step count into the arenas; the reference's own branch never rebuilds the optimizer (dead path)
        from utils import load_checkpoint_file

        ck_model, _values, optim_state, start_epoch, ck_best, _ = load_checkpoint_file(args.continue_from, False, input_size=input_size)
        ck_sd = ck_model.state_dict()
        ref_layout = "mu2_table" not in ck_sd  # a checkpoint of the reference itself: it never kept a table (simple_fhvae.py:51)
        model.load_state_dict(ck_sd, strict=not ref_layout)
        if optim_state is not None:
            if ref_layout:
                # torch.optim.Adam's state over the reference's parameters (the nets, in named_parameters() order): ours has the
                # table in front -> shift by one.  The table's moments start at zero while FusedAdam's ONE step counter continues
                # from the checkpoint: the table's first updates therefore run without bias correction (m and v warm up from 0
                # with the nets' late-step factors ~1: steps of up to ~lr/sqrt(1-beta2) relative size on its first gradients,
                # shrinking over ~1/(1-beta2) steps).  Harmless for a table the reference re-drew from N(0,1) on every forward
                # (simple_fhvae.py:51), and stated here rather than hidden.
                n_have = len(optim_state["param_groups"][0]["params"])
                net_params = [(n, p) for n, p in model.named_parameters() if p.requires_grad and n != "mu2_table"]
                n_nets = len(net_params)
                if n_have != n_nets:
                    raise ValueError("--continue-from: the checkpoint's optimizer holds %d parameters, this model's nets have %d"
                                     % (n_have, n_nets))
                # ... and the SAME parameters in the same order: names from the checkpoint's model, shapes from its moments
                ck_names = [n for n, p in ck_model.named_parameters() if p.requires_grad and n != "mu2_table"]
                if ck_names != [n for n, _ in net_params]:
                    raise ValueError("--continue-from: the checkpoint's parameters %s... are not this model's %s..."
                                     % (ck_names[:3], [n for n, _ in net_params][:3]))
                for k, (n, p) in enumerate(net_params):
                    st = optim_state["state"].get(k)
                    if st is not None and tuple(st["exp_avg"].shape) != tuple(p.shape):
                        raise ValueError("--continue-from: moment %d has shape %s, parameter %s has %s"
                                         % (k, tuple(st["exp_avg"].shape), n, tuple(p.shape)))
                names = [n for n, p in model.named_parameters() if p.requires_grad]
                shift = 1 if names and names[0] == "mu2_table" else 0
                grp = dict(optim_state["param_groups"][0], params=list(range(n_nets + shift)))
                optim_state = {"state": {k + shift: v for k, v in optim_state["state"].items()}, "param_groups": [grp]}
            optimizer.load_state_dict(optim_state)
        if ck_best is not None:
            best_val_lb = float(ck_best)
        ck_raw = torch.load(args.continue_from, map_location="cpu", weights_only=False)
        best_epoch = int(ck_raw.get("best_epoch", start_epoch - 1))  # the patience window continues where it stood
        print(f"resumed from {args.continue_from}: starting at epoch {start_epoch}")

    if real:
        def train_batches():
            return tr_pool.epoch(args.training_batch_size, shuffle=True)

        def dev_batches():
            return dv_pool.epoch(args.dev_batch_size, shuffle=False)

        n_train = len(tr_pool)
    else:
        xtr, itr, ntr = synthetic_split(args.train_segments, T, F, S, args.seed + 1)
        xdv, idv, ndv = synthetic_split(args.dev_segments, T, F, S, args.seed + 2)
        xtr, xdv = xtr.to(device), xdv.to(device)

        def train_batches():
            perm = torch.randperm(xtr.shape[0])
            for s0 in range(0, xtr.shape[0], args.training_batch_size):
                sel = perm[s0:s0 + args.training_batch_size]
                yield itr[sel], xtr[sel], ntr[sel]

        def dev_batches():
            for s0 in range(0, xdv.shape[0], args.dev_batch_size):
                yield idv[s0:s0 + args.dev_batch_size], xdv[s0:s0 + args.dev_batch_size], ndv[s0:s0 + args.dev_batch_size]

        n_train = xtr.shape[0]

    if args.sample_hierarchical:
        # closed-form mu2 from the current encoder (train_model.py:424-436); unlike the reference the result is USED:
        # it initialises the persistent table
        from utils import estimate_mu2_dict

        mu2_dict = estimate_mu2_dict(model, train_batches(), S)
        with torch.no_grad():
            for y, v in mu2_dict.items():
                model.mu2_table[y] = v
        print(f"hierarchical: mu2 re-estimated for {len(mu2_dict)} of {S} sequences")
    if args.exp_dir:
        import os

        os.makedirs(args.exp_dir, exist_ok=True)
        from utils import save_args, save_checkpoint

        save_args(args.exp_dir, args)  # train_model.py:422

    def train_step(idxs, features, nsegs):
        """One iteration of the reference loop body, train_model.py:446-454."""
        optimizer.zero_grad()
        lower_bound, discrim_loss, log_px_z, neg_kld_z1, neg_kld_z2, log_pmu2 = model(features, idxs, S, nsegs)
        loss = loss_function(lower_bound, discrim_loss, args.alpha_dis)
        hb.backward(loss)  # (loss.backward() with a cached seed)
        optimizer.step()
        return loss.detach(), lower_bound.detach()

    graph = None  # --hip-graph: (CUDAGraph, static inputs, static outputs), built on the first full-size batch

    def graph_step(idxs, features, nsegs):
        nonlocal graph
        bsz = args.training_batch_size
        idxs = torch.as_tensor(idxs).to(device=device, dtype=torch.int64)
        nsegs = torch.as_tensor(nsegs).to(device=device, dtype=torch.int64)
        if not args.hip_graph or features.shape[0] != bsz:
            return train_step(idxs, features, nsegs)
        if graph is None:
            st_i, st_x, st_n = idxs.clone(), features.clone(), nsegs.clone()
            # the warm-up steps and the capture must not train: parameters, Adam moments and the step count are put back
            # afterwards, so this batch gets exactly one update (the first replay) like every other batch
            keep = [t.clone() for t in (optimizer.p_arena.flat, optimizer.m, optimizer.v, optimizer.step_dev)]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):  # warm-up outside the capture (lazy initialisations, allocator pools)
                    train_step(st_i, st_x, st_n)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = train_step(st_i, st_x, st_n)
            graph = (g, (st_i, st_x, st_n), outs)
            for t, k in zip((optimizer.p_arena.flat, optimizer.m, optimizer.v, optimizer.step_dev), keep):
                t.copy_(k)
            # capture only RECORDS the kernels: `outs` is uninitialised graph-pool memory until the first replay
            g.replay()
            return outs
        g, (st_i, st_x, st_n), outs = graph
        st_i.copy_(idxs)
        st_x.copy_(features)
        st_n.copy_(nsegs)
        g.replay()
        return outs

    def healthy() -> Optional[int]:
        """One host sync: the sticky device words.  Divergence = NaN lower bound in ANY batch since the start
        (fhvae_loss_fwd's nan_flag; the reference tests every batch on the host, train_model.py:464-466).  Recurrence status
        = a persistent bf16 LSTM launch gave up (all 256 CUs were not co-resident): everything computed since is invalid."""
        if hb.diverged(device):
            print("Training diverged")
            return 2  # sys.exit(2), train_model.py:464-466
        st = hb.lstm_sync_status()
        if st != 0:
            print("a persistent LSTM recurrence launch gave up (status %d): results since are invalid; rerun with "
                  "FHVAE_NO_CLUSTER=1 if the GPU is shared" % st, file=sys.stderr)
            return 3
        return None

    for epoch in range(start_epoch, args.epochs):
        model.train()
        t0 = time.time()
        train_loss = torch.zeros((), device=device)
        nb = 0
        for idxs, features, nsegs in train_batches():
            loss, lower_bound = graph_step(idxs, features, nsegs)
            train_loss += loss
            nb += 1
            if args.check_interval > 0 and nb % args.check_interval == 0:
                rc = healthy()
                if rc is not None:
                    return rc
        rc = healthy()  # end of epoch, and before anything is checkpointed
        if rc is not None:
            return rc
        dt = time.time() - t0
        print(f"====> Train set average loss: {train_loss.item() / nb:.4f}  ({n_train / dt:.0f} segments/s)")
        model.eval()
        lbs = []
        with torch.no_grad():
            for idxs, features, nsegs in dev_batches():
                lbs.append(model(features, idxs, S, nsegs)[0])
        val_lower_bound = torch.cat(lbs)
        print(f"====> Validation set lower bound: {val_lower_bound.mean().item():.4f} "
              f"({val_lower_bound.mean().item() / T:.4f} nats/frame)")
        if check_best(val_lower_bound, best_val_lb):
            best_epoch, best_val_lb = epoch, val_lower_bound.mean().item()
        if args.exp_dir:
            save_checkpoint(model, optimizer, None, {"val_lower_bound": val_lower_bound.mean().item()}, "run", epoch, best_epoch,
                            val_lower_bound.mean().item(), best_val_lb, args.exp_dir, input_size=input_size)
        if check_terminate(epoch, best_epoch, args.patience, args.epochs):
            print("Training terminated!")
            break
    print("Training complete!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
