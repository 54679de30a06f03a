#!/usr/bin/env python3
"""bench.py -- ScalableFHVAE training-step throughput on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = zero_grad -> forward -> loss_function -> backward -> Adam over one synthetic (B,20,80) batch
already resident in HBM (train_model.py:446-454).  Workload at N=1: BASELINE.json configs[1]
(fhvae.FHVAE 2x256 LSTM enc/dec, z1=z2=32, 4.6k-row mu2 table).  value = segments/s over all ranks.
Objective: the intended one (decoder attached, log_qy=-CE): the reference's literal `.detach()`
objective would skip the whole decoder backward, i.e. less work in the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pytorch-scalablefhvae_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

CONFIGS = {
    # configs[0]: the reference's own runnable model (FC SimpleFHVAE; "1-layer LSTM" has no referent) on the HIP path
    "c1": dict(H=128, L=2, D=32, S=100, T=20, F=80, B=250, simple=True, desc="simple_fhvae.SimpleFHVAE 128/128 FC, z1=z2=32, 100-seq mu2 table"),
    # name: (H, layers, D, S, T, F, default per-GPU batch)
    # SURVEY 8d lists C2 at B=256 (the reference's --training-batch-size default, train_model.py:134-137) and at B=2048
    # (its --dev-batch-size default): the headline runs the larger one (the cells are launch-latency-bound at 256) and
    # the B=256 figure is reported next to it in the same JSON line ("alt_batch").
    "c2": dict(H=256, L=2, D=32, S=4600, T=20, F=80, B=2048, alt_B=256, desc="fhvae.FHVAE 2x256 LSTM enc/dec, z1=z2=32, 4.6k-seq mu2 table"),
    "c3": dict(H=256, L=2, D=32, S=28000, T=20, F=80, B=2048, desc="same model, 28k-seq mu2 table, batch 2048"),
    "c4": dict(H=512, L=2, D=32, S=100000, T=20, F=80, B=2048, desc="2x512 LSTM, 100k-seq mu2 table"),
    "c5": dict(H=256, L=2, D=32, S=1000000, T=40, F=80, B=2048, dtype="f32", desc="1M-seq mu2 table, 40-frame segments, fp32"),
}


def lstm_flops_fwd(cfg, B):
    """Algorithmic forward FLOPs of the three LSTM nets: 2*T*sum 4H(I_l+H) per segment."""
    H, L, D, T, F = cfg["H"], cfg["L"], cfg["D"], cfg["T"], cfg["F"]
    per_seg = 0
    for i0 in (F, F + D, 2 * D):
        for l in range(L):
            per_seg += 4 * H * ((i0 if l == 0 else H) + H)
    return 2 * T * per_seg * B


def synth(cfg, B, device, rank):
    """SURVEY 8d synthetic inputs (seeds 1234/1235/1236, offset by rank)."""
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, cfg["T"], cfg["F"], generator=g)
    idx = torch.randint(0, cfg["S"], (B,), generator=torch.Generator().manual_seed(1235 + rank))
    ns = torch.randint(20, 200, (B,), generator=torch.Generator().manual_seed(1236 + rank))
    return x.to(device), idx.to(device), ns.to(device)


def cpu_baseline(cfg, B, budget_s=20.0):
    """The CPU oracle (oracle/ref_cpu.py, kind 'port': the reference's FHVAE is a stub) timed on
    this box's host cores on a bounded sample of the same workload: full training steps at the same batch."""
    from oracle import ref_cpu as R

    H, L, D, S, T, F = (cfg[k] for k in "HLDSTF")
    torch.manual_seed(0)
    m = (R.SimpleFHVAERef(T * F, [H] * L, [H] * L, D, D, [H] * L) if cfg.get("simple")
         else R.FHVAERef(T * F, [H] * L, [H] * L, D, D, [H] * L, seg_len=T))
    table = torch.randn(S, D, requires_grad=True)
    opt = torch.optim.Adam(list(m.parameters()) + [table], lr=1e-3, betas=(0.95, 0.999))
    x = torch.randn(B, T, F)
    idx = torch.randint(0, S, (B,))
    ns = torch.randint(20, 200, (B,))
    e2, e1 = torch.randn(B, D), torch.randn(B, D)
    R.train_step(m, opt, table, x, idx, ns, e2, e1)  # warm-up
    t0, n = time.time(), 0
    while True:
        R.train_step(m, opt, table, x, idx, ns, e2, e1)
        n += 1
        if time.time() - t0 > budget_s or n >= 20:
            break
    dt = time.time() - t0
    return {"value": B * n / dt, "unit": "segments/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d full training steps (fwd+loss+bwd+Adam) of the torch-CPU oracle %s at B=%d, S=%d, fp32"
                      % (n, type(m).__name__, B, S)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32"],
                    help="MFMA operand type of the LSTM nets (default: the config's -- bf16, configs[4] f32; f32 = exact-f32 parity mode)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly instead of replaying the captured hipGraph of the whole step")
    ap.add_argument("--force-dist", action="store_true", help="use the distributed runner even with one rank (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    # stdout carries exactly ONE line (the JSON record): libraries that print banners to fd 1 (RCCL prints its version
    # block there on communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    import hip_binding as hb
    from fhvae import FHVAE
    from hip_optim import FusedAdam
    from train_model import loss_function

    hb.load_library()
    cfg = CONFIGS[args.config]
    args.dtype = args.dtype or cfg.get("dtype", "bf16")
    B = args.batch or cfg["B"]
    H, L, D, S, T, F = (cfg[k] for k in "HLDSTF")
    torch.manual_seed(0)
    if cfg.get("simple"):
        from simple_fhvae import SimpleFHVAE

        model = SimpleFHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, num_seqs=S, reference_compat=False).to(device)
        args.dtype = "f32"
    else:
        model = FHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, seg_len=T, num_seqs=S, reference_compat=False,
                      compute_dtype=args.dtype).to(device)
    with torch.no_grad():
        model.mu2_table.copy_(torch.randn(S, D, generator=torch.Generator().manual_seed(1)))
    if use_dist:
        from dist_shard import DistributedFHVAE

        runner = DistributedFHVAE(model, lr=1e-3, betas=(0.95, 0.999))
    else:
        runner = None
        opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.95, 0.999))
    x, idx, ns = synth(cfg, B, device, rank)

    def step():
        if runner is not None:
            return runner.train_step(x, idx, ns, alpha=10.0)
        opt.zero_grad()
        out = model(x, idx, S, ns)
        loss = loss_function(out[0], out[1], 10.0)
        loss.backward()
        opt.step()
        return loss.detach(), out[0].detach()

    def barrier():
        if use_dist:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    # The whole step (zero_grad, forward, loss, backward, collectives excluded, Adam) is ~400 short launches:
    # capture it once into a hipGraph and replay (single-GPU path; the distributed runner stays eager because
    # RCCL collectives are enqueued from the host between the phases).
    use_graph = not args.no_graph and runner is None
    eager_step = step
    if use_graph:
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)  # capture runs on a side stream by design
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_loss, g_lb = eager_step()

        def step():
            graph.replay()
            return g_loss, g_lb

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, lb = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        import torch.distributed as dist

        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    elbo = (lb.mean() / T).item()
    ok = bool(torch.isfinite(loss).item())
    if hb.lstm_sync_status() != 0:
        raise SystemExit("a persistent LSTM recurrence launch gave up (status %d): results invalid" % hb.lstm_sync_status())

    roof = None
    if not args.no_roofline:  # every rank runs the instrumented steps (they contain collectives); rank 0 reports
        # Instrumented EAGER repeat of the same steps: (1) HIP events (torch's current stream == the launch stream)
        # around every C-ABI call, (2) the library's own per-launch event pairs around every LSTM step-cell
        # launch, each tagged with its algorithmic FLOPs (fhvae_trace_*).  The dominant kernel is the cell kind
        # with the larger total time; achieved = its algorithmic FLOPs / its summed launch durations.
        hb.OP_TIMER.enable()
        hb.cell_trace(True)
        cells = {}
        for _ in range(args.steps):
            eager_step()
            for k, (n, t, f) in hb.cell_trace_collect().items():
                c = cells.get(k, (0, 0.0, 0.0))
                cells[k] = (c[0] + n, c[1] + t, c[2] + f)
        torch.cuda.synchronize()
        hb.cell_trace(False)
        per_op = hb.OP_TIMER.summary()
        hb.OP_TIMER.disable()
        form = hb.LAST_LSTM_FORM["form"]  # which schedule the library took for this shape (fhvae_lstm_form)
        kn = {0: "lstm_%s_step_kernel", 1: "lstm_%s_cluster_kernel", 2: "lstm_%s_ksplit_kernel"}[form]
        names = {0: kn % "fwd", 1: kn % "bwd"}
        if form == 1 and not os.environ.get("FHVAE_NO_LAYERWISE"):
            # rows form: the backward runs one persistent launch per layer (contraction-split variant up to 2048 rows at
            # H = 256: the library's rule in cluster_bwd_layers)
            names[1] = "lstm_bwd_layer_ks_kernel" if H == 256 else "lstm_bwd_layer_kernel"
        if not cells:  # FC model: no LSTM cells to trace
            cells = {0: (1, 1e-9, 0.0)}
        dom = max(cells, key=lambda k: cells[k][1])
        n, t_ms, fl = cells[dom]
        ach = fl / (t_ms * 1e-3) / 1e12
        # dense MFMA peaks (MI355X_MICROARCH.md, matrix cores): bf16 ~2500 TFLOP/s, f32-input 157.3 TFLOP/s
        peak = 2500.0 if args.dtype == "bf16" else 157.3
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):  # HBM bytes per launch from rocprofv3 --pmc passes of this command (tools/pmc_traffic.py)
            try:
                traffic = json.load(open(tf)).get("%s_%s_B%d" % (names[dom], args.dtype, B))
            except Exception:
                traffic = None
        roof = None if rank != 0 else {"bound": "mfma", "kernel": "%s<%s>" % (names[dom], args.dtype), "achieved": ach, "peak": peak,
                "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                "schedule": hb.LSTM_FORMS[form], "launches_per_step": n / args.steps, "avg_launch_us": t_ms / n * 1e3,
                "flops_per_launch": fl / n,
                "cells": {names[k]: {"launches_per_step": v[0] / args.steps, "avg_launch_us": v[1] / v[0] * 1e3,
                                     "tflops": v[2] / (v[1] * 1e-3) / 1e12} for k, v in cells.items()},
                "op_ms_per_step": {k: v[1] / args.steps for k, v in sorted(per_op.items())}}

    alt = None
    if world == 1 and not use_dist and cfg.get("alt_B") and not args.batch and not args.no_roofline:
        # secondary figure at the reference's default training batch (eager launches, same model and optimizer state)
        Ba = cfg["alt_B"]
        xa, ia, na = synth(cfg, Ba, device, rank)
        hb.join_side_stream()

        def step_a():
            opt.zero_grad()
            out = model(xa, ia, S, na)
            l = loss_function(out[0], out[1], 10.0)
            l.backward()
            opt.step()
            return out[0]

        for _ in range(3):
            step_a()
        torch.cuda.synchronize()
        launch_a = "eager"
        run_a = step_a
        if use_graph:  # same method as the headline figure: one captured step, replayed
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step_a()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_a):
                g_lba = step_a()

            def run_a():
                graph_a.replay()
                return g_lba

            launch_a = "hipGraph replay of the whole step"
            for _ in range(3):
                run_a()
            torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(args.steps):
            lba = run_a()
        torch.cuda.synchronize()
        ta = time.perf_counter() - ta
        if hb.lstm_sync_status() != 0:
            raise SystemExit("a persistent LSTM recurrence launch gave up (alt batch): results invalid")
        alt = {"batch": Ba, "value": Ba * args.steps / ta, "unit": "segments/s", "ms_per_step": ta / args.steps * 1e3,
               "launch": launch_a, "elbo_nats_per_frame": (lba.mean() / T).item()}

    if rank == 0:
        rec = {
            "metric": "segments/sec + ELBO (nats/frame), (B,20,80) fbank", "value": world * B * args.steps / dt,
            "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", "elbo_nats_per_frame": elbo, "loss_finite": ok,
            "launch": "hipGraph replay of the whole step" if use_graph else "eager",
            "config": {"workload": "%s: %s; per-GPU batch %d, T=%d, F=%d, full train step (fwd+loss+bwd+Adam), "
                                   "intended objective" % (args.config, cfg["desc"], B, T, F),
                       "global_batch": world * B, "parallelism": "dp%d+mu2-row-shard" % world if use_dist else "single"},
        }
        if roof:
            rec["roofline"] = roof
        if alt:
            rec["alt_batch"] = alt
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(cfg, B)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
    if use_dist:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
