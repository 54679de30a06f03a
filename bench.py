#!/usr/bin/env python3
"""bench.py -- ScalableFHVAE training-step throughput on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run` as a CHILD process, before this process has touched the GPU) and
passes the one JSON line of rank 0 through.

A "step" = zero_grad -> forward -> loss_function -> backward -> Adam over one synthetic (B,20,80) batch
already resident in HBM (train_model.py:446-454).  Workload at N=1: BASELINE.json configs[2], the largest
single-GPU configuration (fhvae.FHVAE 2x256 LSTM enc/dec, z1=z2=32, 28k-row mu2 table, batch 2048, bf16);
configs[1] (4.6k-row table) at batch 2048 and at the reference's default training batch 256, and the headline workload in the
exact-f32 parity mode, are reported beside it as `alt`.  value = segments/s over all ranks.
N > 1: DP over the minibatch + the row-sharded mu2 table (dist_shard.py); beside the c3 headline the `alt` list carries
configs[3] (2x512, 100k rows) and configs[4] (1M rows, T = 40, fp32) -- the workloads north_star's 8-GPU targets are quoted on.
The distributed step is measured EAGERLY first and then as a captured hipGraph; a watchdog prints the eager record and ends the
ranks if the captured form (RCCL collectives inside a graph, never run at N > 1 before the first hardware run) does not come back.
Objective: the intended one (decoder attached, log_qy=-CE): the reference's literal `.detach()`
objective would skip the whole decoder backward, i.e. less work in the timed region.
"""
import argparse
import copy
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pytorch-scalablefhvae_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CONFIGS = {
    # configs[0]: the reference's own runnable model (FC SimpleFHVAE; "1-layer LSTM" has no referent) on the HIP path
    "c1": dict(H=128, L=2, D=32, S=100, T=20, F=80, B=250, simple=True, desc="simple_fhvae.SimpleFHVAE 128/128 FC, z1=z2=32, 100-seq mu2 table"),
    # name: (H, layers, D, S, T, F, default per-GPU batch); batch sizes: the reference's --dev-batch-size 2048 and
    # --training-batch-size 256 defaults (train_model.py:134-137)
    "c2": dict(H=256, L=2, D=32, S=4600, T=20, F=80, B=2048, desc="fhvae.FHVAE 2x256 LSTM enc/dec, z1=z2=32, 4.6k-seq mu2 table"),
    "c3": dict(H=256, L=2, D=32, S=28000, T=20, F=80, B=2048, desc="fhvae.FHVAE 2x256 LSTM enc/dec, z1=z2=32, 28k-seq mu2 table, batch 2048"),
    "c4": dict(H=512, L=2, D=32, S=100000, T=20, F=80, B=2048, desc="2x512 LSTM, 100k-seq mu2 table"),
    "c5": dict(H=256, L=2, D=32, S=1000000, T=40, F=80, B=2048, dtype="f32", desc="1M-seq mu2 table, 40-frame segments, fp32"),
}
#: what the default (no --config) run reports beside the headline: (config, per-GPU batch, dtype or None = the config's)
ALT_RUNS = [("c2", 2048, None), ("c2", 256, None), ("c3", 2048, "f32")]
#: ... and with N > 1 ranks (row-sharded tables: the configurations north_star's multi-GPU targets name)
ALT_RUNS_DIST = [("c4", 2048, None), ("c5", 2048, None)]


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the ranks as a child process tree.  Nothing in
    this process has initialised HIP (no torch.cuda call, no library load): the child processes own the GPUs.  The tree runs
    under a time limit (--launch-timeout); when it expires the whole process group is killed and ONE fresh tree is started with
    --no-dist-graph --no-alt (never a re-exec of a process that touched the GPU); the record's `launch` field says so."""
    import signal
    import socket

    def run(extra, env_extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv + extra
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(env_extra)
        proc = subprocess.Popen(cmd, env=env, start_new_session=True)
        try:
            return proc.wait(timeout=args.launch_timeout if args.launch_timeout > 0 else None)
        except subprocess.TimeoutExpired:
            # the launcher puts its workers into sessions of their own: collect the exact descendants of the tree started above
            # (never a pattern match), then end the launcher's group and every one of them
            pids = []
            try:
                import psutil

                pids = [c.pid for c in psutil.Process(proc.pid).children(recursive=True)]
            except Exception:  # noqa: BLE001
                pass
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            for pid in pids:
                try:
                    os.kill(pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
            proc.wait()
            return None

    rc = run([], {})
    if rc is None:
        print("bench: the %d-rank run did not finish within %d s; starting a fresh one with --no-dist-graph --no-alt"
              % (args.gpus, args.launch_timeout), file=sys.stderr)
        rc = run(["--no-dist-graph", "--no-alt"], {"FHVAE_BENCH_RETRY": "1"})
        if rc is None:
            print("bench: the second attempt did not finish either", file=sys.stderr)
            rc = 124
    return rc


# ---------------------------------------------------------------------------------------------
# watchdog of the multi-rank run: a phase that does not come back (the first captured replay of a step with RCCL collectives
# in it is the candidate) must not lose what was already measured.  One daemon thread per rank; when the armed deadline passes,
# rank 0 prints the record prepared so far (the eager measurement) and every rank ends with os._exit (the main thread is stuck
# inside a HIP / RCCL call: nothing else can end it).
# ---------------------------------------------------------------------------------------------
WATCHDOG = {"deadline": None, "emit": None, "what": "", "started": False}


def watchdog_arm(seconds, what):
    import threading

    WATCHDOG["deadline"], WATCHDOG["what"] = (time.time() + seconds if seconds else None), what
    if WATCHDOG["started"] or not seconds:
        return

    def loop():
        while True:
            time.sleep(1.0)
            d = WATCHDOG["deadline"]
            if d is not None and time.time() > d:
                print("bench: watchdog: '%s' did not finish in time; ending this rank" % WATCHDOG["what"], file=sys.stderr)
                code = 3
                if WATCHDOG["emit"] is not None:
                    try:
                        WATCHDOG["emit"](WATCHDOG["what"])
                        code = 0
                    except Exception as exc:  # noqa: BLE001
                        print("bench: watchdog could not emit the record: %s" % exc, file=sys.stderr)
                sys.stderr.flush()
                os._exit(code)

    WATCHDOG["started"] = True
    threading.Thread(target=loop, daemon=True).start()


def watchdog_disarm():
    WATCHDOG["deadline"] = None


# ---------------------------------------------------------------------------------------------
# synthetic inputs
# ---------------------------------------------------------------------------------------------
def synth_cpu(cfg, B, rank=0, idx_dist="uniform"):
    """x ~ N(0,1) seed 1234, mu_idx seed 1235 (uniform, or bounded Zipf(1.1) over the S rows: p(r) ~ (r+1)^-1.1, the
    gather-collision stress), num_segs in [20,200) seed 1236; seeds offset by the rank."""
    import torch

    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, cfg["T"], cfg["F"], generator=g)
    gi = torch.Generator().manual_seed(1235 + rank)
    if idx_dist == "zipf":
        w = torch.arange(1, cfg["S"] + 1, dtype=torch.float64).pow(-1.1)
        idx = torch.multinomial(w, B, replacement=True, generator=gi)
    else:
        idx = torch.randint(0, cfg["S"], (B,), generator=gi)
    ns = torch.randint(20, 200, (B,), generator=torch.Generator().manual_seed(1236 + rank))
    return x, idx, ns


# ---------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------
def host_cpu_info():
    """One socket's physical cores, clipped to what this process may actually use (affinity mask, cgroup CPU quota)."""
    model, cores = "unknown", {}
    try:
        phys = core = cpu = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu = int(v)
            elif k == "model name":
                model = v
            elif k == "physical id":
                phys = int(v)
            elif k == "core id":
                core = int(v)
            elif not k and cpu is not None:
                cores.setdefault((phys or 0, core if core is not None else cpu), []).append(cpu)
                phys = core = cpu = None
        if cpu is not None:
            cores.setdefault((phys or 0, core if core is not None else cpu), []).append(cpu)
    except OSError:
        pass
    allowed = sorted(os.sched_getaffinity(0))
    sockets = sorted({k[0] for k in cores}) or [0]
    s0 = sockets[0]
    # one logical CPU per physical core of the first socket, restricted to the allowed set
    pin = sorted(min(c for c in cpus if c in allowed) for (s, _), cpus in cores.items()
                 if s == s0 and any(c in allowed for c in cpus))
    if not pin:
        pin = allowed
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        pass
    n = len(pin)
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return {"cpu_model": model, "sockets": len(sockets), "socket_physical_cores": sum(1 for k in cores if k[0] == s0) or len(allowed),
            "allowed_cpus": len(allowed), "cgroup_cpu_quota": quota, "threads": n, "pin": pin[:n]}


def hbm_record(device, B=65536, T=20, F=80, D=32, S=1000000, n_adam=64 * 1024 * 1024):
    """Achieved HBM GB/s of the streaming kernels on the path (north_star: 'achieved HBM GB/s on the gather / NLL kernels'),
    at sizes where a launch is bandwidth- and not latency-sized: K3 lower bound forward / backward (plain and with the bf16 pair
    copy the per-frame head consumes), K4 mu2 gather, Adam, the loader's segment gather.  HIP events on the launch stream
    (torch's current stream: these ops launch there), 10 launches each; bytes = ALGORITHMIC bytes (every operand once).
    -> {name: {"us", "bytes", "gbps", "frac" (of 8 TB/s)}}"""
    import torch

    import hip_binding as hb

    def t(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    out = {}

    def rec(name, sec, nbytes, note=None):
        out[name] = {"us": sec * 1e6, "bytes": nbytes, "gbps": nbytes / sec / 1e9, "frac": nbytes / sec / 8e12}
        if note:
            out[name]["note"] = note

    g = dict(device=device)
    x, xm, xl = torch.randn(T, B, F, **g), torch.randn(T, B, F, **g), torch.randn(T, B, F, **g) * 0.3
    z = [torch.randn(B, D, **g) for _ in range(5)]
    ns = torch.randint(20, 200, (B,), **g)
    lay = (B, T, F, (F, B * F), (F, B * F))
    rec("elbo_fwd", t(lambda: hb.elbo(x, xm, xl, *z, ns, lay, True)), B * (3 * T * F * 4 + 6 * D * 4 + 5 * 4))
    # backward: reads x, x_mu, x_lv and the latents, writes d_x_mu, d_x_lv and the latent gradients
    xm2, xl2 = xm.clone().requires_grad_(True), xl.clone().requires_grad_(True)
    zz = [v.clone().requires_grad_(True) for v in z]
    o = hb.elbo(x, xm2, xl2, *zz, ns, lay, False)
    go = torch.ones(B, **g)
    rec("elbo_bwd", t(lambda: torch.autograd.grad(o[0], [xm2, xl2] + zz, go, retain_graph=True)), B * (5 * T * F * 4 + 10 * D * 4 + 4))
    # ... with mu | logvar side by side (the per-frame head's layout): d_x_mu | d_x_lv in f32 AND once more in bf16 (+ padding)
    pair = torch.randn(T * B, 2 * F, **g)
    pair[:, F:] *= 0.3
    pm, pl = pair[:, :F].requires_grad_(True), pair[:, F:].requires_grad_(True)  # (leaves: no slice-gradient assembly is timed)
    o2 = hb.elbo(x, pm, pl, *zz, ns, lay, False)
    ldg = (2 * F + 63) // 64 * 64
    rec("elbo_bwd_pair", t(lambda: torch.autograd.grad(o2[0], [pm, pl] + zz, go, retain_graph=True)),
        B * (5 * T * F * 4 + T * ldg * 2 + 10 * D * 4 + 4))
    del o, o2, pair, pm, pl, xm2, xl2
    table, idx = torch.randn(S, D, **g), torch.randint(0, S, (B,), **g)
    rec("mu2_gather", t(lambda: hb.raw_gather_rows(table, idx)), B * (2 * D * 4 + 8), "latency-sized even at B = 65536")
    p, gr, m, v = (torch.randn(n_adam, **g) for _ in range(4))
    v.abs_()
    step = torch.ones((), dtype=torch.int32, **g)
    rec("adam", t(lambda: hb.adam_step_(p, gr, m, v, step, 1e-3, 0.95, 0.999, 1e-8)), n_adam * 28)
    del p, gr, m, v
    pool = torch.randn(4_000_000, F, **g)
    st = torch.randint(0, 4_000_000 - T, (B,), **g)
    mean, istd = torch.zeros(F, **g), torch.ones(F, **g)
    rec("segment_gather", t(lambda: hb.segment_gather(pool, st, T, mean, istd)), 2 * B * T * F * 4)
    return out


def cpu_baseline(cfg, dtype, budget_s=45.0, sample_B=256):
    """the CPU oracle (oracle/ref_cpu.py, kind 'port': the reference's FHVAE is a stub; pure
    PyTorch, `torch.nn.LSTM` nets, the (B,S,D) materialisation of simple_fhvae.py:119-121) timed on ONE socket of this
    box: threads = that socket's physical cores (clipped to the cgroup quota / affinity mask), the process pinned to them;
    fp32; same seeds as the GPU run (1234/1235/1236); 5 warm-up + 10 timed full training steps (fewer if they do not fit the
    time budget: the record says how many); median.  Sample: the first `sample_B` segments of the seeded batch against the
    FULL table (the CPU rate is batch-insensitive, SURVEY section 6: 801 vs 1,079 segments/s at 256 vs 2048; at B=2048 the
    (B,S,D) temporaries of this config are 7.3 GB each).
    Matched ELBO: the same steps on the same batches and the same draws run on the GPU in f32 (and in the bench's
    operand dtype) from the same initial weights; ELBO (nats/frame) of the last step's forward is reported for each leg."""
    import torch
    from oracle import ref_cpu as R

    info = host_cpu_info()
    old_aff, old_thr = os.sched_getaffinity(0), torch.get_num_threads()
    try:
        os.sched_setaffinity(0, info["pin"])
    except OSError:
        pass
    torch.set_num_threads(info["threads"])
    H, L, D, S, T, F = (cfg[k] for k in "HLDSTF")
    B = min(sample_B, cfg["B"])
    x, idx, ns = (t[:B] for t in synth_cpu(cfg, cfg["B"]))
    torch.manual_seed(0)
    m = (R.SimpleFHVAERef(T * F, [H] * L, [H] * L, D, D, [H] * L) if cfg.get("simple")
         else R.FHVAERef(T * F, [H] * L, [H] * L, D, D, [H] * L, seg_len=T))
    init_sd = copy.deepcopy(m.state_dict())
    table0 = torch.randn(S, D, generator=torch.Generator().manual_seed(1))
    table = table0.clone().requires_grad_(True)
    opt = torch.optim.Adam(list(m.parameters()) + [table], lr=1e-3, betas=(0.95, 0.999))

    def draws(n):  # per-step reparameterisation draws, the same for every leg
        g = torch.Generator().manual_seed(1237)
        return [(torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)) for _ in range(n)]

    eps = draws(15)
    _, lb = R.train_step(m, opt, table, x, idx, ns, *eps[0])  # first touch (thread pool, allocator): never timed
    t0 = time.perf_counter()
    _, lb = R.train_step(m, opt, table, x, idx, ns, *eps[1])
    t_one = time.perf_counter() - t0
    # 5 warm-up + 10 timed where the budget allows; otherwise as many as fit (at least 2 + 3)
    n_warm, n_timed = 5, 10
    if t_one * 13 > budget_s:
        fit = int(budget_s / t_one)
        n_timed = max(3, min(10, fit - 1))
        n_warm = max(2, min(5, 2 + fit - n_timed))
    times, terms = [], {}
    for k in range(2, n_warm + n_timed):
        t0 = time.perf_counter()
        _, lb = R.train_step(m, opt, table, x, idx, ns, *eps[k], terms=terms if k == n_warm + n_timed - 1 else None)
        dt = time.perf_counter() - t0
        if k >= n_warm:
            times.append(dt)
    n_steps = n_warm + n_timed
    med = statistics.median(times)
    rec = {"value": B / med, "unit": "segments/s", "cores": info["threads"], "threads": info["threads"], "kind": "port",
           "median_ms": med * 1e3, "warmup_steps": n_warm, "timed_steps": len(times), "cpu_model": info["cpu_model"],
           "socket_physical_cores": info["socket_physical_cores"], "sockets": info["sockets"],
           "cgroup_cpu_quota": info["cgroup_cpu_quota"], "pinned_cpus": len(info["pin"]), "dtype": "f32",
           "elbo_nats_per_frame": (lb.mean() / T).item(), "elbo_after_steps": n_steps, "terms": terms,
           "sample": "%d+%d full training steps (fwd+loss+bwd+Adam, median of the timed ones) of the torch-CPU oracle %s on the first "
                     "%d segments of the seeded batch (seeds 1234/1235/1236), full %d-row table with the reference's (B,S,D) "
                     "materialisation, fp32, %d threads pinned to one socket"
                     % (n_warm, len(times), type(m).__name__, B, S, info["threads"])}
    try:
        os.sched_setaffinity(0, old_aff)
    except OSError:
        pass
    torch.set_num_threads(old_thr)

    # the same n_steps steps on the GPU: f32 (parity mode) and the bench's operand dtype, same initial state, batches, draws
    if torch.cuda.is_available():
        from fhvae import FHVAE
        from hip_optim import FusedAdam
        from simple_fhvae import SimpleFHVAE
        from train_model import loss_function

        dev = torch.device("cuda", torch.cuda.current_device())
        xd, idd, nsd = x.to(dev), idx.to(dev), ns.to(dev)
        epd = [(a.to(dev), b.to(dev)) for a, b in eps[:n_steps]]
        for leg in sorted({"f32", dtype}):
            if cfg.get("simple"):
                if leg != "f32":
                    continue
                gm = SimpleFHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, num_seqs=S, reference_compat=False)
            else:
                gm = FHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, seg_len=T, num_seqs=S, reference_compat=False, compute_dtype=leg)
            gm.load_state_dict(init_sd, strict=False)
            with torch.no_grad():
                gm.mu2_table.copy_(table0)
            gm.to(dev)
            gopt = FusedAdam(gm.parameters(), lr=1e-3, betas=(0.95, 0.999))
            for k in range(n_steps):
                gopt.zero_grad()
                out = gm(xd, idd, S, nsd, eps=epd[k])
                loss_function(out[0], out[1], 10.0).backward()
                gopt.step()
            e = (out[0].mean() / T).item()
            # every term of the last forward (batch means, nats per segment) beside the oracle's, and the worst relative difference
            gt = {n: o.detach().mean().item() for n, o in zip(R.TERM_NAMES, out)}
            rec["gpu_%s_terms" % leg] = gt
            rec["gpu_%s_terms_max_rel_diff" % leg] = max(abs(gt[n] - terms[n]) / max(abs(terms[n]), 1.0) for n in terms) if terms else None
            rec["gpu_%s_elbo_nats_per_frame" % leg] = e
            rec["gpu_%s_elbo_rel_diff" % leg] = abs(e - rec["elbo_nats_per_frame"]) / abs(rec["elbo_nats_per_frame"])
            del gm, gopt, out
    return rec


# ---------------------------------------------------------------------------------------------
# one GPU measurement
# ---------------------------------------------------------------------------------------------
def run_gpu(cfg_name, B, dtype, steps, warmup, device, rank, world, use_dist, no_graph, want_roofline, idx_dist, dist_graph=False,
            on_eager=None):
    """on_eager(result): called with the EAGER measurement of a distributed step before its capture into a hipGraph is
    attempted (main arms the watchdog with it)."""
    import torch

    import hip_binding as hb
    from fhvae import FHVAE
    from hip_optim import FusedAdam
    from train_model import loss_function

    cfg = CONFIGS[cfg_name]
    H, L, D, S, T, F = (cfg[k] for k in "HLDSTF")
    torch.manual_seed(0)
    if cfg.get("simple"):
        from simple_fhvae import SimpleFHVAE

        model = SimpleFHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, num_seqs=S, reference_compat=False).to(device)
        dtype = "f32"
    else:
        model = FHVAE(T * F, [H] * L, [H] * L, D, D, [H] * L, seg_len=T, num_seqs=S, reference_compat=False,
                      compute_dtype=dtype).to(device)
    with torch.no_grad():
        model.mu2_table.copy_(torch.randn(S, D, generator=torch.Generator().manual_seed(1)))
    if use_dist:
        import torch.distributed as dist
        from dist_shard import DistributedFHVAE

        runner = DistributedFHVAE(model, lr=1e-3, betas=(0.95, 0.999))
    else:
        runner = None
        opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.95, 0.999))
    x, idx, ns = (t.to(device) for t in synth_cpu(cfg, B, rank, idx_dist))

    def eager_step():
        if runner is not None:
            return runner.train_step(x, idx, ns, alpha=10.0)
        opt.zero_grad()
        out = model(x, idx, S, ns)
        loss = loss_function(out[0], out[1], 10.0)
        hb.backward(loss)
        opt.step()
        return loss.detach(), out[0].detach()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The whole step (zero_grad, forward, loss, backward, Adam) is ~110 short launches: capture it once into a hipGraph and
    # replay.  The distributed runner's step contains RCCL collectives enqueued from the host between the phases: captured
    # too (collectives are graph-capturable; --no-dist-graph runs it eagerly).
    use_graph = (not no_graph) and (runner is None or dist_graph)
    step = eager_step

    def timed(fn):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, lb = fn()
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        return dt, loss, lb

    eager = None
    eager_note = "eager"
    if runner is not None and world > 1:
        # the multi-rank step, eagerly, FIRST: a valid measurement exists before the captured form is tried (and with
        # --no-dist-graph: the place where the ranks agree on the recurrences' status)
        dt, loss, lb = timed(eager_step)
        # A persistent recurrence needs all 256 CUs of its GPU at once; beside a collective kernel that never ran on hardware with
        # this path before, a launch may give up (sticky status).  Every rank must take the same branch: agree on the status, then
        # fall back to the per-step cells (FHVAE_NO_CLUSTER, read by the library per call) and measure again -- a slower valid
        # number instead of none.
        st_all = torch.tensor([hb.lstm_sync_status()], device=device, dtype=torch.int32)
        dist.all_reduce(st_all, op=dist.ReduceOp.MAX)
        if int(st_all.item()) != 0:
            print("bench: a persistent LSTM recurrence launch gave up on some rank (status %d): measuring with the per-step cells"
                  % int(st_all.item()), file=sys.stderr)
            os.environ["FHVAE_NO_CLUSTER"] = "1"
            hb.reset_device_words(device)
            hb.LSTM_WORKSPACES.clear()
            dt, loss, lb = timed(eager_step)
            if hb.lstm_sync_status() != 0:
                raise SystemExit("a persistent LSTM recurrence launch gave up (status %d): results invalid" % hb.lstm_sync_status())
            eager_note = "eager, per-step cells (the persistent recurrences gave up beside the collectives: status %d)" % int(st_all.item())
            use_graph = False  # (the captured form is not tried on top of a fallback)
        eager = {"value": world * B * steps / dt, "ms_per_step": dt / steps * 1e3, "elbo_nats_per_frame": (lb.mean() / T).item(),
                 "loss_finite": bool(torch.isfinite(loss).item()), "launch": eager_note, "dtype": dtype, "batch": B}
        if on_eager is not None:
            on_eager(eager)
    if use_graph:
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)  # capture runs on a side stream by design
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        barrier()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                g_loss, g_lb = eager_step()
        except Exception as exc:  # a step that cannot be captured on this stack (collectives): say so and run it eagerly
            if runner is None:
                raise
            print("bench: capturing the distributed step failed (%s: %s); running it eagerly" % (type(exc).__name__, exc), file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize()
        if use_graph:
            def step():
                graph.replay()
                return g_loss, g_lb

    dt, loss, lb = timed(step)
    elbo = (lb.mean() / T).item()
    ok = bool(torch.isfinite(loss).item())
    if hb.lstm_sync_status() != 0:
        raise SystemExit("a persistent LSTM recurrence launch gave up (status %d): results invalid" % hb.lstm_sync_status())
    res = {"value": world * B * steps / dt, "ms_per_step": dt / steps * 1e3, "elbo_nats_per_frame": elbo, "loss_finite": ok,
           "launch": "hipGraph replay of the whole step" if use_graph else eager_note, "dtype": dtype, "batch": B}
    if eager is not None:
        res["eager"] = {"value": eager["value"], "ms_per_step": eager["ms_per_step"]}
    if os.environ.get("FHVAE_BENCH_RETRY"):
        res["launch"] += " (second attempt: the first %d-rank run did not finish within --launch-timeout)" % world

    if want_roofline:  # every rank runs the instrumented steps (they contain collectives); rank 0 reports
        # Instrumented EAGER repeat of the same steps: (1) HIP events (torch's current stream == the launch stream)
        # around every C-ABI call, (2) the library's own per-launch event pairs around every LSTM step-cell
        # launch, each tagged with its algorithmic FLOPs (fhvae_trace_*).  The dominant kernel is the cell kind
        # with the larger total time; achieved = its algorithmic FLOPs / its summed launch durations.
        hb.OP_TIMER.enable()
        hb.cell_trace(True)
        cells = {}
        for _ in range(steps):
            eager_step()
            for k, (n, t, f) in hb.cell_trace_collect().items():
                c = cells.get(k, (0, 0.0, 0.0))
                cells[k] = (c[0] + n, c[1] + t, c[2] + f)
        torch.cuda.synchronize()
        hb.cell_trace(False)
        per_op = hb.OP_TIMER.summary()
        hb.OP_TIMER.disable()
        form = hb.LAST_LSTM_FORM["form"]  # which schedule the library took for this shape (fhvae_lstm_form)
        names = hb.lstm_kernel_names(form, H)
        if not cells:  # FC model: no LSTM cells to trace
            cells = {0: (1, 1e-9, 0.0)}
        dom = max(cells, key=lambda k: cells[k][1])
        n, t_ms, fl = cells[dom]
        ach = fl / (t_ms * 1e-3) / 1e12
        # dense MFMA peaks (MI355X_MICROARCH.md, matrix cores): bf16 ~2500 TFLOP/s, f32-input 157.3 TFLOP/s
        peak = 2500.0 if dtype == "bf16" else 157.3
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):  # HBM bytes per launch from rocprofv3 --pmc passes of this command (tools/pmc_traffic.py)
            try:
                traffic = json.load(open(tf)).get("%s_%s_B%d" % (names[dom], dtype, B))
            except Exception:
                traffic = None
        hbm_frac = None
        if traffic:  # counter bytes of that kernel / its live average duration / 8 TB/s: how near the HBM roof the kernel is
            traffic = dict(traffic, source="profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command, replayed here: "
                                           "not measured in this run)")
            hbm_frac = traffic["hbm_bytes_per_launch"] / (t_ms / n * 1e-3) / 8e12
        cell_hbm = {}
        try:
            tj = json.load(open(tf)) if os.path.exists(tf) else {}
            for k, v in cells.items():
                e = tj.get("%s_%s_B%d" % (names[k], dtype, B))
                if e:
                    cell_hbm[names[k]] = e["hbm_bytes_per_launch"] / (v[1] / v[0] * 1e-3) / 8e12
        except Exception:
            cell_hbm = {}
        res["roofline"] = {"bound": "mfma", "kernel": "%s<%s>" % (names[dom], dtype), "achieved": ach, "peak": peak,
                           "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "hbm_frac": hbm_frac, "hbm_frac_by_kernel": cell_hbm,
                           "schedule": hb.LSTM_FORMS[form], "launches_per_step": n / steps, "avg_launch_us": t_ms / n * 1e3,
                           "flops_per_launch": fl / n,
                           "cells": {names[k]: {"launches_per_step": v[0] / steps, "avg_launch_us": v[1] / v[0] * 1e3,
                                                "tflops": v[2] / (v[1] * 1e-3) / 1e12} for k, v in cells.items()},
                           "op_ms_per_step": {k: v[1] / steps for k, v in sorted(per_op.items())}}
        if idx_dist == "zipf":
            cnt = torch.bincount(idx.cpu(), minlength=S)
            res["idx"] = {"dist": "bounded Zipf(1.1) over the %d rows" % S, "distinct_rows": int((cnt > 0).sum()),
                          "max_multiplicity": int(cnt.max()),
                          "k4_gather_bwd_ms": per_op.get("fhvae_mu2_gather_bwd", (0, 0.0))[1] / steps,
                          "k5_bwd_ms": per_op.get("fhvae_disc_lse_bwd", (0, 0.0))[1] / steps}
    hb.join_side_stream()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c3 (the largest single-GPU configuration of BASELINE.json), with the c2 figures as `alt`")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32"],
                    help="MFMA operand type of the LSTM nets (default: the config's -- bf16, configs[4] f32; f32 = exact-f32 parity mode)")
    ap.add_argument("--idx", default="uniform", choices=["uniform", "zipf"],
                    help="distribution of mu_idx over the table rows (zipf = bounded Zipf(1.1): the gather-collision stress of SURVEY 8d)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly instead of replaying the captured hipGraph of the whole step")
    ap.add_argument("--force-dist", action="store_true", help="use the distributed runner even with one rank (testing)")
    ap.add_argument("--dist-graph", action="store_true", help="(default now; kept for older command lines)")
    ap.add_argument("--no-hbm", action="store_true", help="skip the roofline.hbm sub-record (achieved GB/s of the streaming kernels)")
    ap.add_argument("--no-dist-graph", action="store_true",
                    help="run the distributed step eagerly instead of replaying its captured hipGraph (collectives included)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-alt", action="store_true")
    ap.add_argument("--launch-timeout", type=int, default=1500,
                    help="self-launched N > 1 run: seconds before the rank tree is killed and ONE fresh tree is started with "
                         "--no-dist-graph --no-alt (0 = no limit)")
    ap.add_argument("--phase-timeout", type=int, default=240,
                    help="N > 1: seconds a captured (hipGraph) phase may take before the watchdog prints the eager record and ends "
                         "the ranks (0 = no watchdog)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    if os.environ.get("FHVAE_BENCH_TEST_SLEEP"):  # tests/test_bench_cpu.py: a rank that never comes back (before anything touches a GPU)
        time.sleep(float(os.environ["FHVAE_BENCH_TEST_SLEEP"]))
    if os.environ.get("FHVAE_BENCH_WATCHDOG"):  # diagnostics: dump every thread's Python stack and exit if the run takes longer
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["FHVAE_BENCH_WATCHDOG"]), exit=True)

    import torch

    # stdout carries exactly ONE line (the JSON record): libraries that print banners to fd 1 (RCCL prints its version
    # block there on communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    import hip_binding as hb

    hb.load_library()
    default_run = args.config is None
    cfg_name = args.config or "c3"
    cfg = CONFIGS[cfg_name]
    dtype = "f32" if cfg.get("simple") else (args.dtype or cfg.get("dtype", "bf16"))
    B = args.batch or cfg["B"]
    T, F = cfg["T"], cfg["F"]
    state = {"main": None, "alts": []}

    def build_record(main_res, note=None):
        rec = {
            "metric": "segments/sec + ELBO (nats/frame), (B,20,80) fbank", "value": main_res["value"],
            "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": main_res["dtype"], "data": "synthetic", "elbo_nats_per_frame": main_res["elbo_nats_per_frame"],
            "loss_finite": main_res["loss_finite"], "launch": main_res["launch"] + (" -- " + note if note else ""),
            "config": {"workload": "%s: %s; per-GPU batch %d, T=%d, F=%d, full train step (fwd+loss+bwd+Adam), "
                                   "intended objective, mu_idx %s" % (cfg_name, cfg["desc"], B, T, F, args.idx),
                       "global_batch": world * B, "parallelism": "dp%d+mu2-row-shard" % world if use_dist else "single"},
        }
        for k in ("roofline", "idx", "eager"):
            if k in main_res:
                rec[k] = main_res[k]
        if state["alts"]:
            rec["alt"] = list(state["alts"])
        return rec

    def emit(rec):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())

    def watchdog_emit(what):  # (watchdog thread) the best record there is: the finished headline, else its eager measurement
        if rank == 0 and state["main"] is not None:
            emit(build_record(state["main"], "watchdog: '%s' did not finish within %d s; this record is what had been measured before it"
                              % (what, args.phase_timeout)))

    guard = world > 1 and args.phase_timeout > 0
    if guard:
        WATCHDOG["emit"] = watchdog_emit

    def on_eager_main(e):
        state["main"] = e
        if guard:
            watchdog_arm(args.phase_timeout, "hipGraph capture + replay of the %s step" % cfg_name)

    main_res = run_gpu(cfg_name, B, dtype, args.steps, args.warmup, device, rank, world, use_dist, args.no_graph,
                       not args.no_roofline, args.idx, use_dist and not args.no_dist_graph, on_eager=on_eager_main)
    watchdog_disarm()
    state["main"] = main_res

    def alt_entry(name, b, r):
        return {"workload": "%s: %s; per-GPU batch %d" % (name, CONFIGS[name]["desc"], b), "value": r["value"],
                "unit": "segments/s", "ms_per_step": r["ms_per_step"], "launch": r["launch"],
                "elbo_nats_per_frame": r["elbo_nats_per_frame"], "dtype": r["dtype"], **({"eager": r["eager"]} if "eager" in r else {})}

    if default_run and world == 1 and not use_dist and not args.batch and not args.no_alt:
        for name, b, dt_alt in ALT_RUNS:  # configs[1] at both reference batch sizes and the headline in f32: same method (graph replay)
            r = run_gpu(name, b, dt_alt or args.dtype or CONFIGS[name].get("dtype", "bf16"), args.steps, 3, device, rank, world, False,
                        args.no_graph, False, args.idx)
            state["alts"].append(alt_entry(name, b, r))
    if default_run and world > 1 and not args.batch and not args.no_alt:
        # the configurations the 8-GPU targets are quoted on (row-sharded 100k / 1M-row tables), same method as the headline;
        # every rank runs them (they contain collectives); a phase that hangs ends the run with what is already measured
        for name, b, dt_alt in ALT_RUNS_DIST:
            if guard:
                watchdog_arm(3 * args.phase_timeout, "alt run %s" % name)
            r = run_gpu(name, b, dt_alt or CONFIGS[name].get("dtype", "bf16"), max(5, args.steps // 2), 3, device, rank, world, True,
                        args.no_graph, False, args.idx, not args.no_dist_graph)
            watchdog_disarm()
            state["alts"].append(alt_entry(name, b, r))

    if rank == 0:
        rec = build_record(main_res)
        if "roofline" in rec and world == 1 and default_run and not args.no_hbm:
            # the HBM-bound kernels of the path beside the MFMA-bound dominant one (~2 s, 5 GB of scratch tensors)
            rec["roofline"]["hbm"] = {"peak_gbps": 8000.0, "kernels": hbm_record(device)}
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(cfg, main_res["dtype"])
        emit(rec)
    if use_dist:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
