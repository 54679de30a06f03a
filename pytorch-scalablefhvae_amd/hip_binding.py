"""ctypes binding of libfhvae_hip.so (C ABI: include/fhvae_hip.h) + autograd wrappers.

This is the only place where Python touches the HIP library.  PyTorch is used for device memory
(the caching allocator, so everything is hipGraph-capturable), the current stream and autograd
bookkeeping; all arithmetic of the hot path runs in the hand-written kernels.  There is NO CPU or
eager-PyTorch fallback: if the library is missing or a tensor is not on a GPU, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfhvae_hip.so")

F32, BF16 = 0, 1
MAX_LAYERS = 4
ABI_VERSION = 11  # FHVAE_ABI_VERSION of include/fhvae_hip.h
#: ``2*exp(pz2_logvar)`` evaluated exactly like simple_fhvae.py:88,:120 (numpy float32 arithmetic)
PZ2_LOGVAR = np.log(0.5 ** 2).astype(np.float32)
INV_TWO_VAR = float(np.float32(1.0) / (np.float32(2.0) * np.exp(PZ2_LOGVAR)))

_vp, _i64, _i32, _f32, _f64 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_double


class LstmDesc(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("L", _i32),
        ("B", _i64), ("T", _i64), ("I", _i64), ("Ic", _i64), ("H", _i64),
        ("x", _vp), ("x_lp", _vp), ("xc", _vp),
        ("w_ih", _vp * MAX_LAYERS), ("w_hh", _vp * MAX_LAYERS),
        ("b_ih", _vp * MAX_LAYERS), ("b_hh", _vp * MAX_LAYERS),
        ("hs", _vp), ("cs", _vp), ("gates", _vp), ("hn", _vp), ("hs_top_f32", _vp), ("pre", _vp), ("lp", _vp),
        ("hn_lp", _vp),
        ("head_w_mu", _vp), ("head_w_lv", _vp), ("head_wl", _vp), ("head_wt", _vp), ("head_D", _i64), ("head_K", _i64), ("head_ldt", _i64),
        ("sticky_status", _vp),
    ]


class LstmBwdDesc(C.Structure):
    _fields_ = [
        ("f", LstmDesc),
        ("d_hs_top", _vp), ("d_hn", _vp),
        ("dgates", _vp), ("dgsum", _vp), ("dc", _vp),
        ("dw_ih", _vp * MAX_LAYERS), ("dw_hh", _vp * MAX_LAYERS),
        ("db_ih", _vp * MAX_LAYERS), ("db_hh", _vp * MAX_LAYERS),
        ("d_xc", _vp), ("phase", _i32), ("ws_below", _vp),
    ]


class ElboDesc(C.Structure):
    _fields_ = [
        ("B", _i64), ("T", _i64), ("F", _i64), ("D1", _i64), ("D2", _i64),
        ("x", _vp), ("x_sb", _i64), ("x_st", _i64),
        ("x_mu", _vp), ("x_lv", _vp), ("xo_sb", _i64), ("xo_st", _i64),
        ("z1_mu", _vp), ("z1_lv", _vp), ("z2_mu", _vp), ("z2_lv", _vp), ("mu2", _vp),
        ("num_segs", _vp), ("nsegs_scalar", _f64),
        ("lower_bound", _vp), ("log_px_z", _vp), ("neg_kld_z1", _vp), ("neg_kld_z2", _vp), ("log_pmu2", _vp),
    ]


class ElboBwdDesc(C.Structure):
    _fields_ = [
        ("f", ElboDesc),
        ("g_lower_bound", _vp), ("g_log_px_z", _vp), ("g_neg_kld_z1", _vp), ("g_neg_kld_z2", _vp), ("g_log_pmu2", _vp),
        ("reference_detach", _i32),
        ("d_x_mu", _vp), ("d_x_lv", _vp), ("d_z1_mu", _vp), ("d_z1_lv", _vp), ("d_z2_mu", _vp), ("d_z2_lv", _vp),
        ("d_mu2", _vp),
        ("d_x_pair_lp", _vp), ("ld_pair", _i64), ("d_x_colsum", _vp),
    ]


class WgradDesc(C.Structure):
    _fields_ = [("a", _vp), ("lda", _i64), ("a_col0", _i64), ("b", _vp), ("ldb", _i64), ("c", _vp), ("ldc", _i64),
                ("M", _i64), ("N", _i64), ("K", _i64)]


#: every symbol include/fhvae_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "fhvae_abi_version": (C.c_int, []),
    "fhvae_strerror": (C.c_char_p, [C.c_int]),
    "fhvae_linear_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i64, C.c_int, C.c_int, _vp]),
    "fhvae_linear_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64,
                                   _i64, C.c_int, C.c_int, _vp]),
    "fhvae_gauss_head_reparam_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_int,
                                               _vp]),
    "fhvae_gauss_head_pair_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "fhvae_gauss_reparam_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "fhvae_gauss_head_bwd": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "fhvae_head_pair_weights": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "fhvae_gauss_reparam_pair_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "fhvae_gauss_reparam_bwd_pair": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _vp]),
    "fhvae_elbo_colsum_rows": (_i64, [_i64]),
    "fhvae_gauss_head_bwd_pair": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "fhvae_loss_fwd": (C.c_int, [_vp, _vp, _f32, _vp, _i64, _vp, _vp]),
    "fhvae_loss_bwd": (C.c_int, [_vp, _f32, _vp, _vp, _i64, _vp]),
    "fhvae_lstm_lp_bytes": (_i64, [C.POINTER(LstmDesc)]),
    "fhvae_lstm_form": (C.c_int, [C.POINTER(LstmDesc)]),
    "fhvae_lstm_layout_id": (C.c_int, [C.POINTER(LstmDesc)]),
    "fhvae_lstm_pre_elems": (_i64, [C.POINTER(LstmDesc)]),
    "fhvae_lstm_ws_below_elems": (_i64, [C.POINTER(LstmDesc)]),
    "fhvae_lstm_seq_fwd": (C.c_int, [C.POINTER(LstmDesc), _vp]),
    "fhvae_lstm_seq_bwd": (C.c_int, [C.POINTER(LstmBwdDesc), _vp]),
    "fhvae_lstm_param_grads_multi": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp]),
    "fhvae_wgrad_desc_ok": (C.c_int, [_vp]),
    "fhvae_wgrad_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp]),
    "fhvae_wgrad_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp]),
    "fhvae_proj_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "fhvae_mu2_gather_fwd": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp]),
    "fhvae_mu2_gather_bwd": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _f32, _vp]),
    "fhvae_shard_pack": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "fhvae_shard_unpack": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "fhvae_disc_merge_partials": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "fhvae_shard_bwd_pack": (C.c_int, [_vp, _f32, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "fhvae_shard_bwd_unpack": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp]),
    "fhvae_disc_ce_mean": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _i64, _vp]),
    "fhvae_elbo_fwd": (C.c_int, [C.POINTER(ElboDesc), _vp]),
    "fhvae_elbo_bwd": (C.c_int, [C.POINTER(ElboBwdDesc), _vp]),
    "fhvae_disc_lse_ws_bytes": (_i64, [_i64, _i64]),
    "fhvae_disc_lse_bwd_ws_bytes": (_i64, [_i64, _i64, _i64]),
    "fhvae_disc_lse_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _f32, _vp, _i64, _i64, _i64, C.c_int, _vp]),
    "fhvae_disc_lse_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, C.c_int, _vp]),
    "fhvae_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, C.c_int, _vp, _vp]),
    "fhvae_segment_gather": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "fhvae_mu2_accumulate": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "fhvae_mu2_finalize": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f32, _vp]),
    "fhvae_trace_enable": (C.c_int, [C.c_int]),
    "fhvae_trace_collect": (_i64, [_vp, _vp, _vp, _i64]),
    "fhvae_to_time_major": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_int, _vp]),
    "fhvae_cast_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
}

_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            "libfhvae_hip.so is not built (%s). Run `python pytorch-scalablefhvae_amd/build_ext.py` "
            "(or __graft_entry__.build()). There is no CPU fallback." % path
        )
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.fhvae_abi_version() != ABI_VERSION:
        raise RuntimeError("libfhvae_hip.so ABI version mismatch")
    _lib = lib
    return lib


# ---------------------------------------------------------------------------------------------
# second stream: the weight-gradient contractions of a net (wide GEMMs, many workgroups) run there, under the
# next net's recurrence (<= 128 workgroups per launch, latency-bound).  Used only when every parameter has a
# gradient sink (hip_optim.FusedAdam): nothing downstream reads those gradients before the optimizer, which joins.
# ---------------------------------------------------------------------------------------------
# MEASURED (MI355X, c2, B=256, bf16): enabling it made the step SLOWER (2.61 -> 2.98 ms under graph replay): the
# concurrent GEMM workgroups take L2/LDS bandwidth from the latency-bound cells on the critical path.  Off by default.
_SIDE = {"stream": None, "pending": False, "keep": [], "enabled": False}


#: optional callback(sinks) fired when a net's backward recurrence has been enqueued (its parameter gradients queued behind it, or
#: about to run): the data-parallel wrapper flushes the queue / starts the gradient all-reduce of the finished buckets there
LSTM_BWD_REC_HOOK = {"fn": None}


# ---------------------------------------------------------------------------------------------
# Deferred parameter gradients.  The weight gradients of an LSTM net depend only on what its backward recurrence left
# (dgates) and on saved states; nothing on the rest of the backward pass depends on them.  When every parameter has a
# gradient sink (hip_optim.FusedAdam's flat arena) the backward of a net therefore only runs its recurrence and QUEUES its
# parameter-gradient phase; the optimizer (step / flat_grad / zero_grad) flushes the queue as ONE grouped call
# (fhvae_lstm_param_grads_multi): the 12 long weight-gradient contractions of the three nets become one launch of whole
# 256x256 tiles with a few K slices instead of 12 launches of 512 split-K workgroups each.
# A caller that reads `param.grad` between backward() and the optimizer must call flush_param_grads() first.
# ---------------------------------------------------------------------------------------------
# "extra": the heads' weight-gradient contractions (WgradDesc, keep-alive tensors), which ride in the same grouped launch
_DEFER = {"enabled": not os.environ.get("FHVAE_NO_DEFER"), "pending": [], "extra": []}


def flush_param_grads():
    """Run the queued parameter-gradient phases (current stream).  No-op when nothing is queued."""
    pend, extra = _DEFER["pending"], _DEFER["extra"]
    if not pend and not extra:
        return
    lib = load_library()
    n, nx = len(pend), len(extra)
    arr = (C.POINTER(LstmBwdDesc) * n)(*[C.pointer(bd) for bd, _ in pend]) if n else None
    xs = (WgradDesc * nx)(*[x for x, _ in extra]) if nx else None
    with _Timed("fhvae_lstm_param_grads_multi"):
        _check(lib.fhvae_lstm_param_grads_multi(arr, n, xs, nx, _stream()), "fhvae_lstm_param_grads_multi")
    pend.clear()  # (the caching allocator keeps the released buffers ordered behind this stream's queued work)
    extra.clear()


def flush_param_grads_except_last():
    """Flush every queued parameter-gradient phase but the most recent one (the net whose recurrence was just enqueued)."""
    pend = _DEFER["pending"]
    if len(pend) <= 1:
        return
    last = pend.pop()
    flush_param_grads()
    pend.append(last)


def set_defer_param_grads(on: bool):
    flush_param_grads()
    _DEFER["enabled"] = bool(on)


def side_stream():
    if _SIDE["stream"] is None:
        _SIDE["stream"] = torch.cuda.Stream()
    return _SIDE["stream"]


def join_side_stream():
    """Make the current stream wait for the side stream's pending work (call before reading the gradient arena)."""
    if _SIDE["pending"]:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["pending"] = False
        _SIDE["keep"].clear()  # safe to release: the current stream is now ordered after their last use


def set_side_stream_enabled(on: bool):
    join_side_stream()
    _SIDE["enabled"] = bool(on)


class _OpTimer:
    """Optional HIP-event timing of every C-ABI call (bench.py's roofline leg).  Events are recorded on
    torch's current stream, which is the stream the kernels are enqueued on."""

    def __init__(self):
        self.on = False
        self.pairs = []

    def enable(self):
        self.on, self.pairs = True, []

    def disable(self):
        self.on = False

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name, e0):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.pairs.append((name, e0, e1))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.pairs:
            n, t = out.get(name, (0, 0.0))
            out[name] = (n + 1, t + e0.elapsed_time(e1))
        return out


OP_TIMER = _OpTimer()


def cell_trace(enable: bool):
    """Turn the in-library per-launch event trace of the LSTM step cells on/off (clears it)."""
    _check(load_library().fhvae_trace_enable(int(enable)), "fhvae_trace_enable")


def cell_trace_collect(cap: int = 65536):
    """-> dict kind -> (launches, total_ms, total_flops); kinds: 0 forward cell, 1 backward cell."""
    lib = load_library()
    ms = (C.c_float * cap)()
    kind = (C.c_int32 * cap)()
    fl = (C.c_double * cap)()
    n = int(lib.fhvae_trace_collect(ms, kind, fl, cap))
    out = {}
    for i in range(n):
        c, t, f = out.get(kind[i], (0, 0.0, 0.0))
        out[kind[i]] = (c + 1, t + ms[i], f + fl[i])
    return out


class _Timed:
    """`with _Timed("fhvae_xxx"):` around a library call."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.e0 = OP_TIMER.start() if OP_TIMER.on else None

    def __exit__(self, *exc):
        if self.e0 is not None:
            OP_TIMER.stop(self.name, self.e0)


def _check(code: int, what: str):
    if code != 0:
        msg = load_library().fhvae_strerror(code).decode()
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg, code))


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "fhvae HIP op called with a CPU tensor: the hot path has no CPU fallback "
                "(move the model and inputs to a MI355X device)"
            )


def _sink(t):
    """Gradient sink of a parameter: hip_optim.FusedAdam registers `param._fh_grad` (a view of its flat
    gradient arena).  Backward kernels then ACCUMULATE straight into it and the Function returns None for
    that input: no zero-filled temporary, no `param.grad += g` pass per parameter per step."""
    return getattr(t, "_fh_grad", None) if t is not None else None


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError("fhvae HIP ops take float32 tensors (got %s)" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# raw (no-autograd) calls used by the Functions below and by tests
# ---------------------------------------------------------------------------------------------
def raw_linear_fwd(x, w, b, relu=False):
    lib = load_library()
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    with _Timed("fhvae_linear_fwd"):
        _check(lib.fhvae_linear_fwd(_p(x), x.stride(0), _p(w), w.stride(0), _p(b), _p(y), N, None, M, K, N, int(relu), F32,
                                    _stream()), "fhvae_linear_fwd")
    return y


def raw_linear_bwd(x, w, y, dy, relu, need_dx=True, need_dw=True, need_db=True, dx_out=None, dw_sink=None, db_sink=None):
    """dw_sink / db_sink: existing buffers to accumulate into (returned dw / db are then None)."""
    lib = load_library()
    M, K = x.shape
    N = w.shape[0]
    dev = x.device
    masked = torch.empty(M, N, device=dev, dtype=torch.float32) if relu else None
    acc = dx_out is not None
    dx = dx_out if acc else (torch.empty(M, K, device=dev, dtype=torch.float32) if need_dx else None)
    dw = dw_sink if dw_sink is not None else (torch.zeros(N, K, device=dev, dtype=torch.float32) if need_dw else None)
    db = db_sink if db_sink is not None else (torch.zeros(N, device=dev, dtype=torch.float32) if need_db else None)
    with _Timed("fhvae_linear_bwd"):
        _check(lib.fhvae_linear_bwd(_p(x), x.stride(0), _p(w), w.stride(0), _p(y), N if y is not None else 0, _p(dy),
                                    dy.stride(0), _p(masked), _p(dx), K, _p(dw), K, _p(db), M, K, N, int(relu), int(acc),
                                    _stream()), "fhvae_linear_bwd")
    return dx, dw, db


# ---------------------------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) -- nn.Linear(+ReLU), simple_fhvae.py:127-134."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        _need_gpu(x, w, b)
        ctx.sinks = (_sink(w), _sink(b))
        x, w, b = _f32c(x), _f32c(w), _f32c(b)
        y = raw_linear_fwd(x, w, b, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _f32c(dy)
        dx, dw, db = raw_linear_bwd(x, w, y, dy, ctx.relu, need_dx=ctx.needs_input_grad[0],
                                    need_dw=ctx.needs_input_grad[1], need_db=ctx.needs_input_grad[2],
                                    dw_sink=ctx.sinks[0], db_sink=ctx.sinks[1])
        # a gradient that went straight into the optimizer's sink must NOT be handed to autograd as well (AccumulateGrad
        # would add the sink to itself: the FC model's weight gradients were doubled under FusedAdam)
        return dx, (None if ctx.sinks[0] is not None else dw), (None if ctx.sinks[1] is not None else db), None


def linear(x, w, b, relu=False):
    return _Linear.apply(x, w, b, bool(relu))


class _GaussHead(torch.autograd.Function):
    """(mu, logvar, sample) = GaussianLayer(h) with an injected eps -- simple_fhvae.py:193-216."""

    @staticmethod
    def forward(ctx, h, w_mu, b_mu, w_lv, b_lv, eps):
        _need_gpu(h, w_mu, b_mu, w_lv, b_lv, eps)
        ctx.set_materialize_grads(False)  # unused outputs arrive as None in backward (no zero tensors, no reads of them)
        lib = load_library()
        ctx.sinks = tuple(_sink(t) for t in (w_mu, b_mu, w_lv, b_lv))
        h, w_mu, b_mu, w_lv, b_lv = _f32c(h), _f32c(w_mu), _f32c(b_mu), _f32c(w_lv), _f32c(b_lv)
        M, K = h.shape
        D = w_mu.shape[0]
        mu = torch.empty(M, D, device=h.device, dtype=torch.float32)
        lv = torch.empty_like(mu)
        if eps is not None:
            eps = _f32c(eps)
            smp = torch.empty_like(mu)
        else:
            smp = None
        with _Timed("fhvae_gauss_head_reparam_fwd"):
            _check(lib.fhvae_gauss_head_reparam_fwd(_p(h), h.stride(0), _p(w_mu), _p(w_lv), _p(b_mu), _p(b_lv), _p(eps), _p(mu),
                                                    _p(lv), _p(smp), M, K, D, F32, _stream()), "fhvae_gauss_head_reparam_fwd")
        ctx.save_for_backward(h, w_mu, w_lv, eps, lv)
        if smp is None:
            smp = mu.new_empty(())  # placeholder output (value never read), never differentiable
            ctx.mark_non_differentiable(smp)
        return mu, lv, smp

    @staticmethod
    def backward(ctx, d_mu, d_lv, d_s):
        lib = load_library()
        h, w_mu, w_lv, eps, lv = ctx.saved_tensors
        if eps is None:
            d_s = None
        M, D = lv.shape
        K = h.shape[1]
        d_mu = _f32c(d_mu) if d_mu is not None else None
        d_lv = _f32c(d_lv) if d_lv is not None else None
        d_s = _f32c(d_s) if d_s is not None else None
        need_dh = ctx.needs_input_grad[0]
        sk = ctx.sinks
        # gradient accumulators: the optimizer's sinks when present (FusedAdam arena), else fresh zeros returned to autograd
        outs = [k if k is not None else torch.zeros(shape, device=h.device, dtype=torch.float32)
                for k, shape in zip(sk, ((D, K), (D,), (D, K), (D,)))]
        dw_mu, db_mu, dw_lv, db_lv = outs
        g_ws = torch.empty(M, 2 * D, device=h.device, dtype=torch.float32)
        dh = torch.empty(M, K, device=h.device, dtype=torch.float32) if need_dh else None
        with _Timed("fhvae_gauss_head_bwd"):
            _check(lib.fhvae_gauss_head_bwd(_p(h), h.stride(0), _p(w_mu), _p(w_lv), _p(d_mu), _p(d_lv), _p(d_s), _p(eps), _p(lv),
                                            _p(g_ws), _p(dh), K, _p(dw_mu), _p(dw_lv), _p(db_mu), _p(db_lv), M, K, D, _stream()),
                   "fhvae_gauss_head_bwd")
        dw_mu, db_mu, dw_lv, db_lv = (None if k is not None else o for k, o in zip(sk, outs))
        return dh, dw_mu, db_mu, dw_lv, db_lv, None


# The upstream gradient of a head's (mu | logvar) pair buffer can arrive ready-made: fhvae_elbo_bwd writes the decoder-output
# gradients a second time as the bf16 operand of the head's backward contractions, with their column sums (the bias
# gradients).  _Elbo.backward leaves them here with the f32 gradient buffer it returns to autograd; the head's backward takes
# them only when the gradients autograd hands it ARE views of that very tensor (identity of the base, same address and strides)
# AND the buffer has not been written since the kernel filled it (version counter: an in-place hook such as g.mul_() keeps the
# address but bumps the version).  One entry: the latest; every _Elbo.backward clears it first, so nothing stale survives a
# backward that produced no side copy or a head that declined it.
_PAIR_GRAD: dict = {}
PAIR_SIDE = {"enabled": True, "used": 0}  # tests: switch the ready-made operand off / count how often a head took it


def _pair_ld(D: int) -> int:
    return (2 * D + 63) // 64 * 64  # whole 64-k stages of the projection kernel; the padding is zero


class _GaussHeadLp(torch.autograd.Function):
    """_GaussHead with bf16 MFMA operands: `h` (f32) only carries the gradient, the contractions read `h_lp` (the same
    values in bf16) and stacked bf16 copies of the two weight matrices: mu | logvar come out of ONE projection side by side,
    and the backward is one projection (dh), one weight-gradient launch and a column sum over one bf16 operand
    g = [g_mu | g_lv].  Outputs, gradients and accumulation are f32."""

    @staticmethod
    def forward(ctx, h, h_lp, w_mu, b_mu, w_lv, b_lv, eps, shadows=None):
        """shadows: (wl, wt) already produced for these weights in this step (the LSTM forward's operand-cast launch,
        fhvae_lstm_desc.head_*), else one fhvae_head_pair_weights launch here."""
        _need_gpu(h, h_lp, w_mu, b_mu, w_lv, b_lv, eps)
        lib = load_library()
        ctx.set_materialize_grads(False)
        ctx.sinks = tuple(_sink(t) for t in (w_mu, b_mu, w_lv, b_lv))
        w_mu, b_mu, w_lv, b_lv = _f32c(w_mu), _f32c(b_mu), _f32c(w_lv), _f32c(b_lv)
        M, K = h_lp.shape
        D = w_mu.shape[0]
        assert h_lp.dtype == torch.bfloat16 and h_lp.is_contiguous() and h.shape == h_lp.shape
        dev = h_lp.device
        ldg = _pair_ld(D)
        if shadows is not None and tuple(shadows[0].shape) == (2 * D, K) and tuple(shadows[1].shape) == (K, ldg):
            wl, wt = shadows
        else:
            wl = torch.empty(2 * D, K, device=dev, dtype=torch.bfloat16)   # [W_mu; W_lv]
            wt = torch.empty(K, ldg, device=dev, dtype=torch.bfloat16)     # [W_mu^T | W_lv^T | 0]: the backward's operand
            _check(lib.fhvae_head_pair_weights(_p(w_mu), _p(w_lv), _p(wl), _p(wt), ldg, D, K, _stream()), "fhvae_head_pair_weights")
        out = torch.empty(M, 2 * D, device=dev, dtype=torch.float32)
        with _Timed("fhvae_gauss_head_reparam_fwd"):
            _check(lib.fhvae_gauss_head_pair_fwd(_p(h_lp), K, _p(wl), _p(b_mu), _p(b_lv), _p(out), 2 * D, M, K, D, _stream()),
                   "fhvae_gauss_head_pair_fwd")
            if eps is not None:  # a sampling head (the latents): contiguous mu, logvar and the sample from one more small launch
                eps = _f32c(eps)
                mu, lv, smp = (torch.empty(M, D, device=dev, dtype=torch.float32) for _ in range(3))
                _check(lib.fhvae_gauss_reparam_pair_fwd(_p(out), 2 * D, _p(eps), _p(smp), _p(mu), _p(lv), M, D, _stream()),
                       "fhvae_gauss_reparam_pair_fwd")
            else:  # the per-frame head: mu | logvar stay side by side (the lower bound reads them in place)
                mu, lv = out[:, :D], out[:, D:]
                smp = mu.new_empty(())  # placeholder (value never read)
                ctx.mark_non_differentiable(smp)
        ctx.save_for_backward(h_lp, wt, eps, lv)
        return mu, lv, smp

    @staticmethod
    def backward(ctx, d_mu, d_lv, d_s):
        lib = load_library()
        h_lp, wt, eps, lv = ctx.saved_tensors
        if eps is None:
            d_s = None
        M, D = lv.shape
        K = h_lp.shape[1]
        ldg = wt.shape[1]
        need_dh = ctx.needs_input_grad[0]
        sk = ctx.sinks
        dev = h_lp.device
        outs = [k if k is not None else torch.zeros(shape, device=dev, dtype=torch.float32)
                for k, shape in zip(sk, ((D, K), (D,), (D, K), (D,)))]
        dh = torch.empty(M, K, device=dev, dtype=torch.float32) if need_dh else None
        g_lp = colsum = None
        db_done = False
        ready = _PAIR_GRAD.get("latest")
        if (ready is not None and d_s is None and d_mu is not None and d_lv is not None and ready[0].data_ptr() == d_mu.data_ptr()
                and d_mu._base is ready[0] and d_lv._base is ready[0] and ready[0]._version == ready[3]
                and d_lv.data_ptr() == d_mu.data_ptr() + 4 * D and d_mu.shape == (M, D) and d_lv.shape == (M, D)
                and d_mu.stride() == (2 * D, 1) and d_lv.stride() == (2 * D, 1) and ready[1].shape == (M, ldg)):
            g_lp, colsum = ready[1], ready[2]
            _PAIR_GRAD.clear()
            PAIR_SIDE["used"] += 1
        with _Timed("fhvae_gauss_head_bwd"):
            if g_lp is None:
                d_mu = _f32c(d_mu) if d_mu is not None else None
                d_lv = _f32c(d_lv) if d_lv is not None else None
                if d_s is not None and not (d_s.dtype == torch.float32 and d_s.dim() == 2 and d_s.stride(1) == 1 and d_s.stride(0) >= D):
                    d_s = _f32c(d_s)  # (a column slice of the next net's input gradient -- cat's backward -- goes in as it is)
                g_lp = torch.empty(M, ldg, device=dev, dtype=torch.bfloat16)
                # (the two bias gradients = column sums of g_lp come out of the same launch)
                _check(lib.fhvae_gauss_reparam_bwd_pair(_p(d_mu), _p(d_lv), _p(d_s), d_s.stride(0) if d_s is not None else D, _p(eps), _p(lv), lv.stride(0), _p(g_lp), ldg,
                                                        _p(outs[1]), _p(outs[3]), M, D, _stream()), "fhvae_gauss_reparam_bwd_pair")
                db_done = True
            # every parameter has a gradient sink: the two weight-gradient contractions join the nets' grouped launch
            xs = None
            if _DEFER["enabled"] and not _SIDE["enabled"] and all(k is not None for k in sk):
                xs = [WgradDesc(_p(g_lp) + 2 * i * D, ldg, i * D, _p(h_lp), K, _p(outs[2 * i]), K, D, K, M) for i in range(2)]
                if not all(lib.fhvae_wgrad_desc_ok(C.byref(x)) for x in xs):
                    xs = None
            dws = (None, None) if xs is not None else (outs[0], outs[2])
            _check(lib.fhvae_gauss_head_bwd_pair(_p(h_lp), K, _p(wt), ldg, _p(g_lp), ldg, _p(colsum), colsum.shape[0] if colsum is not None else 0, _p(dh), K,
                                                 _p(dws[0]), _p(dws[1]),
                                                 None if db_done else _p(outs[1]), None if db_done else _p(outs[3]), M, K, D, _stream()),
                   "fhvae_gauss_head_bwd_pair")
            if xs is not None:
                _DEFER["extra"].extend((x, (g_lp, h_lp, outs[0], outs[2])) for x in xs)
        dw_mu, db_mu, dw_lv, db_lv = (None if k is not None else o for k, o in zip(sk, outs))
        return dh, None, dw_mu, db_mu, dw_lv, db_lv, None, None


def head_shadow_shapes(w_mu):
    """Shapes of the stacked bf16 operands (wl, wt) of a bf16 head with weights like w_mu [D,K], or None if it takes the f32 path."""
    D, K = w_mu.shape
    if K % 8 or D % 8:
        return None
    return (2 * D, K), (K, _pair_ld(D))


def gauss_head(h, w_mu, b_mu, w_lv, b_lv, eps, h_lp=None, shadows=None):
    """h_lp: optional bf16 copy of h -> the contractions run on bf16 MFMA operands (K and D multiples of 8).
    shadows: the head's stacked bf16 weights if something earlier in the step already made them (lstm_seq(head=...))."""
    if h_lp is not None and h.shape[1] % 8 == 0 and w_mu.shape[0] % 8 == 0:
        mu, lv, smp = _GaussHeadLp.apply(h, h_lp, w_mu, b_mu, w_lv, b_lv, eps, shadows)
    else:
        mu, lv, smp = _GaussHead.apply(h, w_mu, b_mu, w_lv, b_lv, eps)
    return mu, lv, (smp if eps is not None else None)


def cast_bf16(t: torch.Tensor) -> torch.Tensor:
    """bf16 copy of a contiguous f32 matrix (one launch; no gradient): the operand of a bf16 head."""
    _need_gpu(t)
    t = _f32c(t.detach())
    out = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
    R = t.shape[0]
    _check(load_library().fhvae_cast_bf16(_p(t), _p(out), None, R, t.numel() // R, _stream()), "fhvae_cast_bf16")
    return out


def to_time_major(x: torch.Tensor, with_bf16: bool = False):
    """(B,T,F) -> (T,B,F) copy (input data: no gradient).  with_bf16: also return the bf16 copy the bf16 LSTM nets
    consume (one pass produces both, and both encoders share it)."""
    _need_gpu(x)
    lib = load_library()
    x = _f32c(x.detach())
    B, T, F_ = x.shape
    out = torch.empty(T, B, F_, device=x.device, dtype=torch.float32)
    if with_bf16:
        lp = torch.empty(T, B, F_, device=x.device, dtype=torch.bfloat16)
        with _Timed("fhvae_to_time_major"):
            _check(lib.fhvae_to_time_major(_p(x), _p(lp), _p(out), B, T, F_, BF16, _stream()), "fhvae_to_time_major")
        out._fh_lp = lp  # rides along with the f32 tensor; hip_binding.lstm_seq picks it up
        return out
    with _Timed("fhvae_to_time_major"):
        _check(lib.fhvae_to_time_major(_p(x), _p(out), None, B, T, F_, F32, _stream()), "fhvae_to_time_major")
    return out


def _fill_lstm_desc(d, dtype, dims, x_tm, xc, params, x_lp=None):
    L, B, T, I, Ic, H = dims
    d.dtype, d.L, d.B, d.T, d.I, d.Ic, d.H = dtype, L, B, T, I, Ic, H
    d.x, d.xc = _p(x_tm), _p(xc)
    d.x_lp = _p(x_lp) if dtype == BF16 else None
    for l in range(L):
        d.w_ih[l], d.w_hh[l], d.b_ih[l], d.b_hh[l] = (_p(params[4 * l + k]) for k in range(4))
    d.sticky_status = _p(_device_words(params[0].device)) if dtype == BF16 else None


# Two sticky int32 device words per GPU, never cleared by the library: [0] = OR of the status words of every persistent
# recurrence launch that gave up (fhvae_lstm_desc.sticky_status), [1] = the divergence flag of fhvae_loss_fwd (nan_flag).
# The training loops read them once per epoch / before a checkpoint instead of synchronising every batch.
_DEVICE_WORDS: dict = {}


def _device_words(device) -> torch.Tensor:
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    w = _DEVICE_WORDS.get(key)
    if w is None:
        w = _DEVICE_WORDS[key] = torch.zeros(4, device=torch.device("cuda", key), dtype=torch.int32)
    return w


def diverged(device=None) -> bool:
    """True once any fhvae_loss_fwd on this device has seen a NaN lower bound (train_model.py:464-466's test; one sync)."""
    w = _device_words(device if device is not None else torch.cuda.current_device())
    return bool(w[1].item() != 0)


def reset_device_words(device=None):
    _device_words(device if device is not None else torch.cuda.current_device()).zero_()


# the most recent bf16 workspaces (fhvae_lstm_desc.lp): word 0 of each is the status of the persistent recurrence
# kernels that last ran on it (include/fhvae_hip.h, FHVAE_LSTM_SYNC_BYTES)
LSTM_WORKSPACES: list = []


LSTM_FORMS = {0: "one launch per wavefront step", 1: "persistent cluster kernel (waves split rows)",
              2: "persistent cluster kernel (waves split the contraction)"}
LAST_LSTM_FORM = {"form": 0}


def lstm_kernel_names(form: int, H: int) -> dict:
    """{0: forward, 1: backward} kernel names of the recurrence schedule `form` (fhvae_lstm_form) as rocprofv3 lists them."""
    kn = {0: "lstm_%s_step_kernel", 1: "lstm_%s_cluster_kernel", 2: "lstm_%s_ksplit_kernel"}[form]
    names = {0: kn % "fwd", 1: kn % "bwd"}
    if form == 1:
        # rows form: the backward runs one persistent launch per layer (H = 256: partial-dh exchange, lstm_bwd_rs.hip), the
        # forward of a two-layer H = 256 net with register-stationary weights (lstm_fwd_wr.hip)
        names[1] = "lstm_bwd_layer_rs_kernel" if H == 256 and not os.environ.get("FHVAE_NO_RS") else "lstm_bwd_layer_kernel"
        if H == 256 and not os.environ.get("FHVAE_NO_RS") and not os.environ.get("FHVAE_NO_FWD_WR"):
            names[0] = "lstm_fwd_wr_kernel"
    return names


def lstm_sync_status() -> int:
    """OR of the sticky per-device status words (every earlier forward folds its workspace's previous status into them) and
    of the status words of the most recent bf16 LSTM workspaces (synchronises).  Non-zero: a persistent recurrence launch
    gave up at some point since the process started (bounded spin expired / unexpected workgroup placement): every result
    computed since is suspect."""
    st = 0
    for w in _DEVICE_WORDS.values():
        st |= int(w[0].item())
    for lp in LSTM_WORKSPACES:
        st |= int(lp[:4].view(torch.int32).item())
    return st


class _LstmSeq(torch.autograd.Function):
    """Multi-layer LSTM over the whole segment (K1).  Inputs: x_tm (T,B,I) or None, xc (B,Ic) or None,
    then per layer w_ih, w_hh, b_ih, b_hh (all f32).  Outputs: hs_top (T,B,H) f32 and hn (B, L*H).
    dtype = F32 (exact-f32 MFMA) or BF16 (bf16 MFMA operands, f32 accumulate / cell state)."""

    @staticmethod
    def forward(ctx, x_tm, xc, T, dtype, top, head, *params):
        """top: 2 = f32 top-layer h_t is an output (default); 1 (bf16 only) = the returned f32 tensor only ROUTES the
        gradient, its values are undefined and the data is its `_fh_lp` bf16 twin; 0 = no per-step output at all (only hn).
        head: None or (w_mu, w_lv) of the Gaussian head that reads this net's output (bf16 mode): its stacked bf16 operands are
        made by the forward's operand-cast launch and ride on the outputs as `_fh_head`."""
        lib = load_library()
        ctx.set_materialize_grads(False)  # encoders use only hn, the decoder only hs_top: the other gradient stays None
        _need_gpu(x_tm, xc, *params)
        L = len(params) // 4
        assert len(params) == 4 * L and 1 <= L <= MAX_LAYERS
        ctx.sinks = [_sink(p) for p in params]
        params = [_f32c(p) for p in params]
        H = params[1].shape[1]
        x_lp = getattr(x_tm, "_fh_lp", None) if x_tm is not None else None
        x_tm = _f32c(x_tm) if x_tm is not None else None
        xc = _f32c(xc) if xc is not None else None
        I = x_tm.shape[2] if x_tm is not None else 0
        Ic = xc.shape[1] if xc is not None else 0
        B = x_tm.shape[1] if x_tm is not None else xc.shape[0]
        if x_tm is not None:
            assert x_tm.shape[0] == T
        assert params[0].shape == (4 * H, I + Ic), (params[0].shape, H, I, Ic)
        dev = params[0].device
        f32 = dict(device=dev, dtype=torch.float32)
        bf = dtype == BF16
        hs = torch.empty(L, T, B, H, device=dev, dtype=torch.bfloat16 if bf else torch.float32)
        cs = torch.empty(L, T, B, H, **f32)
        gates = torch.empty(L, T, B, 4 * H, device=dev, dtype=hs.dtype)
        hn = torch.empty(B, L * H, **f32)
        # the latent head's bf16 operand: only where the final states ARE the output (the encoders: top == 0)
        hn_lp = torch.empty(B, L * H, device=dev, dtype=torch.bfloat16) if (bf and top == 0) else None
        if not bf:
            top = 2
        hs_top = torch.empty(T, B, H, **f32) if (bf and top != 0) else None
        d = LstmDesc()
        dims = (L, B, T, I, Ic, H)
        _fill_lstm_desc(d, dtype, dims, x_tm, xc, params, x_lp)
        # workspace: bf16 operand copies + sync block (bf16 mode) / transposed f32 weights for the backward cells (f32 mode)
        lp = torch.empty(int(lib.fhvae_lstm_lp_bytes(C.byref(d))), device=dev, dtype=torch.uint8)
        if bf:
            LSTM_WORKSPACES.append(lp)
            del LSTM_WORKSPACES[:-16]
        d.lp = _p(lp)
        d.hs, d.cs, d.gates, d.hn, d.hs_top_f32, d.lp = _p(hs), _p(cs), _p(gates), _p(hn), _p(hs_top if top == 2 else None), _p(lp)
        d.hn_lp = _p(hn_lp)
        shadows = None
        if bf and head is not None and head_shadow_shapes(head[0]) is not None:
            hw_mu, hw_lv = _f32c(head[0].detach()), _f32c(head[1].detach())
            (sl, st_) = head_shadow_shapes(hw_mu)
            shadows = (torch.empty(sl, device=dev, dtype=torch.bfloat16), torch.empty(st_, device=dev, dtype=torch.bfloat16))
            d.head_w_mu, d.head_w_lv, d.head_wl, d.head_wt = _p(hw_mu), _p(hw_lv), _p(shadows[0]), _p(shadows[1])
            d.head_D, d.head_K, d.head_ldt = hw_mu.shape[0], hw_mu.shape[1], st_[1]
        # layer-0 input projection workspace: (T,B,4H) only for the schedules that read it (168 MB per net at B = 2048, H = 256)
        pre = torch.empty(max(1, int(lib.fhvae_lstm_pre_elems(C.byref(d)))), **f32)
        d.pre = _p(pre)
        LAST_LSTM_FORM["form"] = int(lib.fhvae_lstm_form(C.byref(d)))
        with _Timed("fhvae_lstm_seq_fwd"):
            _check(lib.fhvae_lstm_seq_fwd(C.byref(d), _stream()), "fhvae_lstm_seq_fwd")
        ctx.dims, ctx.dtype = dims, dtype
        ctx.x_lp = x_lp
        ctx.layout_id = int(lib.fhvae_lstm_layout_id(C.byref(d)))  # the schedule this forward took (see backward)
        ctx.save_for_backward(x_tm, xc, hs, cs, gates, lp, *params)
        if hn_lp is not None:
            hn._fh_lp = hn_lp  # the same values in bf16, written by the forward itself (fhvae_lstm_desc.hn_lp)
        hn._fh_head = shadows
        if top == 0:
            out = hn.new_empty(())  # placeholder (value never read): this net's per-step states are not an output
            ctx.mark_non_differentiable(out)
            return out, hn
        out = hs_top if bf else hs[L - 1]
        if bf:
            out._fh_lp = hs[L - 1]  # the same values in bf16 (what the recurrence itself consumed): operand of a bf16 head
        out._fh_head = shadows
        return out, hn

    @staticmethod
    def backward(ctx, d_hs_top, d_hn):
        lib = load_library()
        L, B, T, I, Ic, H = ctx.dims
        x_tm, xc, hs, cs, gates, lp = ctx.saved_tensors[:6]
        params = ctx.saved_tensors[6:]
        dev = hs.device
        f32 = dict(device=dev, dtype=torch.float32)
        if d_hs_top is None and d_hn is None:
            return (None,) * (6 + 4 * L)
        d_hs_top = _f32c(d_hs_top) if d_hs_top is not None else None
        d_hn = _f32c(d_hn) if d_hn is not None else None
        bd = LstmBwdDesc()
        d = bd.f
        _fill_lstm_desc(d, ctx.dtype, ctx.dims, x_tm, xc, params, ctx.x_lp)
        pre = torch.empty(1, **f32)  # not used by the backward, must be non-NULL
        d.hs, d.cs, d.gates, d.hn, d.hs_top_f32, d.pre, d.lp = _p(hs), _p(cs), _p(gates), None, None, _p(pre), _p(lp)
        dgates = torch.empty(L, T, B, 4 * H, device=dev, dtype=hs.dtype)
        dgsum = torch.empty(B, 4 * H, **f32) if Ic > 0 else None
        dc = torch.empty(L, B, H, **f32)
        grads = [sk if sk is not None else torch.zeros_like(p) for p, sk in zip(params, ctx.sinks)]
        d_xc = torch.empty(B, Ic, **f32) if (Ic > 0 and ctx.needs_input_grad[1]) else None
        bd.d_hs_top, bd.d_hn = _p(d_hs_top), _p(d_hn)
        bd.dgates, bd.dgsum, bd.dc = _p(dgates), _p(dgsum), _p(dc)
        for l in range(L):
            bd.dw_ih[l], bd.dw_hh[l], bd.db_ih[l], bd.db_hh[l] = (_p(grads[4 * l + k]) for k in range(4))
        bd.d_xc = _p(d_xc)
        if int(lib.fhvae_lstm_layout_id(C.byref(d))) != ctx.layout_id:
            raise RuntimeError("the LSTM schedule changed between this net's forward and its backward (an FHVAE_* switch was flipped "
                               "in between?): the saved gates / workspaces are in the forward's layout")
        n_below = int(lib.fhvae_lstm_ws_below_elems(C.byref(d)))
        ws_below = torch.empty(n_below, **f32) if n_below > 0 else None
        bd.ws_below = _p(ws_below)
        if (_DEFER["enabled"] and not _SIDE["enabled"] and all(sk is not None for sk in ctx.sinks)):
            # recurrence now; the parameter gradients with those of the other nets at the optimizer (flush_param_grads)
            bd.phase = 1
            with _Timed("fhvae_lstm_seq_bwd"):
                _check(lib.fhvae_lstm_seq_bwd(C.byref(bd), _stream()), "fhvae_lstm_seq_bwd")
            _DEFER["pending"].append((bd, (x_tm, xc, hs, cs, gates, lp, pre, dgates, dgsum, dc, d_hs_top, d_hn, params, ctx.x_lp)))
            if LSTM_BWD_REC_HOOK["fn"] is not None:  # the distributed runner decides when to flush the queue and start collectives
                LSTM_BWD_REC_HOOK["fn"](ctx.sinks)
            return (None, d_xc, None, None, None, None, *[None] * len(params))
        if _SIDE["enabled"] and all(sk is not None for sk in ctx.sinks):
            # recurrence on this stream; the weight-gradient contractions on the side stream, joined by the optimizer
            bd.phase = 1
            with _Timed("fhvae_lstm_seq_bwd"):
                _check(lib.fhvae_lstm_seq_bwd(C.byref(bd), _stream()), "fhvae_lstm_seq_bwd")
            main, side = torch.cuda.current_stream(), side_stream()
            side.wait_stream(main)
            bd.phase = 2
            with torch.cuda.stream(side):
                with _Timed("fhvae_lstm_seq_bwd(param grads, side stream)"):
                    _check(lib.fhvae_lstm_seq_bwd(C.byref(bd), _stream()), "fhvae_lstm_seq_bwd")
            _SIDE["pending"] = True
            _SIDE["keep"].append((x_tm, xc, hs, cs, gates, lp, dgates, dgsum, dc, d_hs_top, d_hn, params))
        elif LSTM_BWD_REC_HOOK["fn"] is not None:
            # recurrence, hook (the distributed runner starts collectives that may overlap the weight-gradient
            # contractions but must not overlap a persistent recurrence kernel), then the parameter gradients
            bd.phase = 1
            with _Timed("fhvae_lstm_seq_bwd"):
                _check(lib.fhvae_lstm_seq_bwd(C.byref(bd), _stream()), "fhvae_lstm_seq_bwd")
            LSTM_BWD_REC_HOOK["fn"](ctx.sinks)
            bd.phase = 2
            with _Timed("fhvae_lstm_seq_bwd(param grads)"):
                _check(lib.fhvae_lstm_seq_bwd(C.byref(bd), _stream()), "fhvae_lstm_seq_bwd")
        else:
            bd.phase = 0
            with _Timed("fhvae_lstm_seq_bwd"):
                _check(lib.fhvae_lstm_seq_bwd(C.byref(bd), _stream()), "fhvae_lstm_seq_bwd")
        return (None, d_xc, None, None, None, None, *[None if sk is not None else g for g, sk in zip(grads, ctx.sinks)])


def lstm_seq(x_tm, xc, T, params: Sequence[torch.Tensor], dtype: int = F32, top: int = 2, head=None):
    """top (bf16 only; see _LstmSeq.forward): 2 = f32 top-layer states are written and returned; 1 = the returned f32
    tensor carries the gradient only (values undefined, data in its `_fh_lp`); 0 = only the final states are wanted.
    head: (w_mu, w_lv) of the Gaussian head behind this net, or None (see _LstmSeq.forward)."""
    return _LstmSeq.apply(x_tm, xc, int(T), int(dtype), int(top), head, *params)


def raw_gather_rows(table, idx, idx_offset=0):
    """rows = table[idx - idx_offset], zeros where the row is outside the table (a shard's view)."""
    lib = load_library()
    S, D = table.shape
    B = idx.shape[0]
    out = torch.empty(B, D, device=table.device, dtype=torch.float32)
    with _Timed("fhvae_mu2_gather_fwd"):
        _check(lib.fhvae_mu2_gather_fwd(_p(table), _p(idx), idx_offset, _p(out), B, S, D, None, _stream()),
               "fhvae_mu2_gather_fwd")
    return out


def raw_scatter_rows_(dtable, drows, idx, idx_offset=0, scale=1.0):
    """dtable[idx - idx_offset] += scale * drows (rows outside the table skipped)."""
    lib = load_library()
    S, D = dtable.shape
    with _Timed("fhvae_mu2_gather_bwd"):
        _check(lib.fhvae_mu2_gather_bwd(_p(drows), _p(idx), idx_offset, _p(dtable), idx.shape[0], S, D, float(scale),
                                        _stream()), "fhvae_mu2_gather_bwd")


class _Mu2Gather(torch.autograd.Function):
    """mu2 = table[idx] -- simple_fhvae.py:53."""

    @staticmethod
    def forward(ctx, table, idx):
        _need_gpu(table, idx)
        ctx.sink = _sink(table)
        table = _f32c(table)
        out = raw_gather_rows(table, idx)
        ctx.save_for_backward(idx)
        ctx.shape = tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, dmu2):
        (idx,) = ctx.saved_tensors
        dmu2 = _f32c(dmu2)
        dt = ctx.sink if ctx.sink is not None else torch.zeros(ctx.shape, device=dmu2.device, dtype=torch.float32)
        raw_scatter_rows_(dt, dmu2, idx)
        return (None if ctx.sink is not None else dt), None


def mu2_gather(table, idx):
    return _Mu2Gather.apply(table, idx)


def _fill_elbo_desc(d: ElboDesc, x, x_strides, x_mu, x_lv, xo_strides, z1_mu, z1_lv, z2_mu, z2_lv, mu2, num_segs, B, T, F_):
    d.B, d.T, d.F, d.D1, d.D2 = B, T, F_, z1_mu.shape[1], z2_mu.shape[1]
    d.x, (d.x_sb, d.x_st) = _p(x), x_strides
    d.x_mu, d.x_lv, (d.xo_sb, d.xo_st) = _p(x_mu), _p(x_lv), xo_strides
    d.z1_mu, d.z1_lv, d.z2_mu, d.z2_lv, d.mu2 = _p(z1_mu), _p(z1_lv), _p(z2_mu), _p(z2_lv), _p(mu2)
    if isinstance(num_segs, torch.Tensor):
        d.num_segs, d.nsegs_scalar = _p(num_segs), 0.0
    else:
        d.num_segs, d.nsegs_scalar = None, float(num_segs)


class _Elbo(torch.autograd.Function):
    """Fused lower bound (K3) -- simple_fhvae.py:105-116.  `layout` = (B,T,F, x strides, x_mu strides)."""

    @staticmethod
    def forward(ctx, x, x_mu, x_lv, z1_mu, z1_lv, z2_mu, z2_lv, mu2, num_segs, layout, reference_detach):
        _need_gpu(x, x_mu, x_lv, z1_mu, z1_lv, z2_mu, z2_lv, mu2)
        ctx.set_materialize_grads(False)  # the loss uses lower_bound only; the four reporting outputs carry no gradient
        lib = load_library()
        B, T, F_, xs, xos = layout
        # mu | logvar side by side in one (rows, 2F) buffer (the per-frame head's single projection): read in place through the
        # row stride, and the backward writes its two gradients side by side too
        pair = (x_mu.dim() == 2 and x_lv.dim() == 2 and x_mu.dtype == torch.float32 and x_lv.dtype == torch.float32
                and x_mu.stride(1) == 1 and x_lv.stride(1) == 1 and x_mu.stride(0) == 2 * F_ and x_lv.stride(0) == 2 * F_
                and x_lv.data_ptr() - x_mu.data_ptr() == 4 * F_ and xos == (F_, B * F_))
        if pair:
            xos = (2 * F_, B * 2 * F_)
            layout = (B, T, F_, xs, xos)
        ts = [t if (pair and i in (1, 2)) else _f32c(t) for i, t in enumerate((x, x_mu, x_lv, z1_mu, z1_lv, z2_mu, z2_lv, mu2))]
        ctx.pair = pair
        if isinstance(num_segs, torch.Tensor):
            num_segs = num_segs.to(device=ts[0].device, dtype=torch.int64).contiguous()
        outs = [torch.empty(B, device=ts[0].device, dtype=torch.float32) for _ in range(5)]
        d = ElboDesc()
        _fill_elbo_desc(d, ts[0], xs, ts[1], ts[2], xos, *ts[3:], num_segs, B, T, F_)
        d.lower_bound, d.log_px_z, d.neg_kld_z1, d.neg_kld_z2, d.log_pmu2 = (_p(o) for o in outs)
        with _Timed("fhvae_elbo_fwd"):
            _check(lib.fhvae_elbo_fwd(C.byref(d), _stream()), "fhvae_elbo_fwd")
        ctx.layout, ctx.detach = layout, bool(reference_detach)
        ctx.nsegs = num_segs
        ctx.save_for_backward(*ts)
        if reference_detach:  # reference: log_px_z and log_pmu2 carry no gradient (simple_fhvae.py:107,114)
            ctx.mark_non_differentiable(outs[1], outs[4])
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_lb, g_px, g_k1, g_k2, g_pm):
        lib = load_library()
        ts = ctx.saved_tensors
        B, T, F_, xs, xos = ctx.layout
        _PAIR_GRAD.clear()
        bd = ElboBwdDesc()
        _fill_elbo_desc(bd.f, ts[0], xs, ts[1], ts[2], xos, *ts[3:], ctx.nsegs, B, T, F_)
        gs = [_f32c(g) if g is not None else None for g in (g_lb, g_px, g_k1, g_k2, g_pm)]
        if ctx.detach:
            gs[1] = gs[4] = None
        bd.g_lower_bound, bd.g_log_px_z, bd.g_neg_kld_z1, bd.g_neg_kld_z2, bd.g_log_pmu2 = (_p(g) for g in gs)
        bd.reference_detach = int(ctx.detach)
        need_x = not ctx.detach
        side = None
        if need_x and ctx.pair:
            dbuf = torch.empty(ts[1].shape[0], 2 * F_, device=ts[1].device, dtype=torch.float32)
            d_xmu, d_xlv = dbuf[:, :F_], dbuf[:, F_:]
            # time-major rows, whole float4 groups: the kernel also leaves the bf16 operand + column sums for the head's backward
            if PAIR_SIDE["enabled"] and xs == (F_, B * F_) and F_ % 4 == 0 and F_ <= 256 and ts[0].data_ptr() % 16 == 0 and ts[1].data_ptr() % 16 == 0:
                ldg = _pair_ld(F_)
                side = (dbuf, torch.empty(T * B, ldg, device=dbuf.device, dtype=torch.bfloat16),
                        torch.empty(int(lib.fhvae_elbo_colsum_rows(B)), 2 * F_, device=dbuf.device, dtype=torch.float32))
                bd.d_x_pair_lp, bd.ld_pair, bd.d_x_colsum = _p(side[1]), ldg, _p(side[2])
        else:
            d_xmu = torch.empty_like(ts[1]) if need_x else None
            d_xlv = torch.empty_like(ts[2]) if need_x else None
        dz = [torch.empty_like(t) for t in ts[3:8]]
        bd.d_x_mu, bd.d_x_lv = _p(d_xmu), _p(d_xlv)
        bd.d_z1_mu, bd.d_z1_lv, bd.d_z2_mu, bd.d_z2_lv, bd.d_mu2 = (_p(t) for t in dz)
        with _Timed("fhvae_elbo_bwd"):
            _check(lib.fhvae_elbo_bwd(C.byref(bd), _stream()), "fhvae_elbo_bwd")
        if side is not None:
            _PAIR_GRAD["latest"] = side + (dbuf._version,)  # (recorded after the kernel call: ctypes writes do not count)
        return (None, d_xmu, d_xlv, *dz, None, None, None)


def elbo(x, x_mu, x_lv, z1_mu, z1_lv, z2_mu, z2_lv, mu2, num_segs, layout, reference_detach):
    if reference_detach:  # simple_fhvae.py:114: the decoder outputs are detached -> its graph is not reached
        x_mu, x_lv = x_mu.detach(), x_lv.detach()
    return _Elbo.apply(x, x_mu, x_lv, z1_mu, z1_lv, z2_mu, z2_lv, mu2, num_segs, layout, bool(reference_detach))


class _FusedLoss(torch.autograd.Function):
    """loss = -(mean(lower_bound) + alpha * log_qy), train_model.py:243-251, one launch each way."""

    @staticmethod
    def forward(ctx, lower_bound, log_qy, alpha):
        _need_gpu(lower_bound, log_qy)
        lib = load_library()
        lb, qy = _f32c(lower_bound), _f32c(log_qy)
        out = torch.empty((), device=lb.device, dtype=torch.float32)
        with _Timed("fhvae_loss_fwd"):
            _check(lib.fhvae_loss_fwd(_p(lb), _p(qy), float(alpha), _p(out), lb.numel(), _device_words(lb.device)[1:].data_ptr(),
                                      _stream()), "fhvae_loss_fwd")
        ctx.alpha, ctx.B = float(alpha), lb.numel()
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        g = _f32c(g)
        d_lb = torch.empty(ctx.B, device=g.device, dtype=torch.float32)
        d_qy = torch.empty((), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        with _Timed("fhvae_loss_bwd"):
            _check(lib.fhvae_loss_bwd(_p(g), ctx.alpha, _p(d_lb), _p(d_qy), ctx.B, _stream()), "fhvae_loss_bwd")
        return d_lb, d_qy, None


def fused_loss(lower_bound, log_qy, alpha):
    return _FusedLoss.apply(lower_bound, log_qy, float(alpha))


def raw_disc_fwd(q, table, idx, row0=0, want_ce=True, lp=False, out3=None, ce_scale=1.0):
    """lp: the bf16 compute mode's kernels (split-operand bf16 MFMA) where they apply (D = 32, B*S >= 65536).
    out3: an optional (3, B) f32 buffer that receives (row_max, row_sumexp, tgt_logit) as its rows.
    ce_scale: the returned scalar is ce_scale * CE (-1: the intended objective's log_qy without a negation launch)."""
    lib = load_library()
    B, D = q.shape
    S = table.shape[0]
    dev = q.device
    ws = torch.empty(max(int(lib.fhvae_disc_lse_ws_bytes(B, S)), 8), device=dev, dtype=torch.uint8)
    rmax, rsum, tgt = (out3[0], out3[1], out3[2]) if out3 is not None else (torch.empty(B, device=dev, dtype=torch.float32) for _ in range(3))
    ce = torch.empty((), device=dev, dtype=torch.float32) if want_ce else None
    with _Timed("fhvae_disc_lse_fwd"):
        _check(lib.fhvae_disc_lse_fwd(_p(q), _p(table), _p(idx), row0, INV_TWO_VAR, _p(rmax), _p(rsum), _p(tgt), _p(ce), float(ce_scale),
                                      _p(ws), B, S, D, BF16 if lp else F32, _stream()), "fhvae_disc_lse_fwd")
    return rmax, rsum, tgt, ce


#: default workspace size of the one-pass K5 backward: None = the library's recommendation; 0 = two passes (tests / tools)
DISC_BWD_WS = {"bytes": None}


def raw_disc_bwd(q, table, idx, rmax, rsum, g_scale, g_mul, row0=0, need_dq=True, need_dt=True, dt_sink=None, lp=False, ws_bytes=None):
    """ws_bytes: size of the one-pass form's workspace (default: the library's recommendation, capped at 1.5 GiB; a smaller
    workspace makes the kernels take the queries in groups; 0 = two passes)."""
    lib = load_library()
    B, D = q.shape
    S = table.shape[0]
    dq = torch.empty(B, D, device=q.device, dtype=torch.float32) if need_dq else None
    dt = dt_sink if dt_sink is not None else (torch.zeros(S, D, device=q.device, dtype=torch.float32) if need_dt else None)
    # workspace of the one-pass form (both gradients from one recomputation of the logits)
    if ws_bytes is None:
        ws_bytes = DISC_BWD_WS["bytes"]
    nws = int(lib.fhvae_disc_lse_bwd_ws_bytes(B, S, D)) if ws_bytes is None else int(ws_bytes)
    ws = torch.empty(nws, device=q.device, dtype=torch.uint8) if (need_dq and dt is not None and nws > 0) else None
    with _Timed("fhvae_disc_lse_bwd"):
        _check(lib.fhvae_disc_lse_bwd(_p(q), _p(table), _p(idx), row0, INV_TWO_VAR, _p(rmax), _p(rsum), _p(g_scale), float(g_mul),
                                      _p(dq), _p(dt), _p(ws), nws if ws is not None else 0, B, S, D, BF16 if lp else F32, _stream()),
               "fhvae_disc_lse_bwd")
    return dq, (None if dt_sink is not None else dt)


def shard_pack(q, idx):
    """[q | int32 bits of idx] rows (fhvae_shard_pack)."""
    lib = load_library()
    B, D = q.shape
    out = torch.empty(B, D + 1, device=q.device, dtype=torch.float32)
    _check(lib.fhvae_shard_pack(_p(q), _p(idx), _p(out), B, D, _stream()), "fhvae_shard_pack")
    return out


def shard_unpack(pk):
    lib = load_library()
    N, D = pk.shape[0], pk.shape[1] - 1
    q = torch.empty(N, D, device=pk.device, dtype=torch.float32)
    idx = torch.empty(N, device=pk.device, dtype=torch.int64)
    _check(lib.fhvae_shard_unpack(_p(pk), _p(q), _p(idx), N, D, _stream()), "fhvae_shard_unpack")
    return q, idx


def disc_merge_partials(parts):
    """parts (W, 3, N) -> (row_max, row_sumexp, tgt_logit) of the whole table (fhvae_disc_merge_partials)."""
    lib = load_library()
    W, _, N = parts.shape
    m, s, t = (torch.empty(N, device=parts.device, dtype=torch.float32) for _ in range(3))
    _check(lib.fhvae_disc_merge_partials(_p(parts), _p(m), _p(s), _p(t), W, N, _stream()), "fhvae_disc_merge_partials")
    return m, s, t


def shard_bwd_pack(dq_all, dq_scale, dmu2_local, own0, n_all, D):
    lib = load_library()
    ref = dq_all if dq_all is not None else dmu2_local
    out = torch.empty(n_all, 2 * D, device=ref.device, dtype=torch.float32)
    n_own = dmu2_local.shape[0] if dmu2_local is not None else 0
    _check(lib.fhvae_shard_bwd_pack(_p(dq_all), float(dq_scale), _p(dmu2_local), own0, n_own, _p(out), n_all, D, _stream()),
           "fhvae_shard_bwd_pack")
    return out


def shard_bwd_unpack(buf, own0, n_own, want_dq=True, want_dmu2=True):
    lib = load_library()
    N, D = buf.shape[0], buf.shape[1] // 2
    dq = torch.empty(n_own, D, device=buf.device, dtype=torch.float32) if want_dq else None
    dm = torch.empty(N, D, device=buf.device, dtype=torch.float32) if want_dmu2 else None
    _check(lib.fhvae_shard_bwd_unpack(_p(buf), own0, n_own, _p(dq), _p(dm), N, D, _stream()), "fhvae_shard_bwd_unpack")
    return dq, dm


def raw_disc_ce_mean(m, s, tgt, scale=1.0):
    lib = load_library()
    ce = torch.empty((), device=m.device, dtype=torch.float32)
    with _Timed("fhvae_disc_ce_mean"):
        _check(lib.fhvae_disc_ce_mean(_p(m), _p(s), _p(tgt), _p(ce), float(scale), m.numel(), _stream()), "fhvae_disc_ce_mean")
    return ce


class _DiscLse(torch.autograd.Function):
    """log_qy = CrossEntropy(-(q - table)^2 / (2 var), idx), mean over the batch (K5) --
    simple_fhvae.py:119-122, without the (B,S,D) temporaries."""

    @staticmethod
    def forward(ctx, q, table, idx, lp, sign):
        _need_gpu(q, table, idx)
        ctx.sink = _sink(table)
        ctx.lp = bool(lp)
        ctx.sign = float(sign)
        q, table = _f32c(q), _f32c(table)
        rmax, rsum, _, ce = raw_disc_fwd(q, table, idx, lp=ctx.lp, ce_scale=ctx.sign)
        ctx.save_for_backward(q, table, idx, rmax, rsum)
        return ce

    @staticmethod
    def backward(ctx, g):
        q, table, idx, rmax, rsum = ctx.saved_tensors
        g = _f32c(g).reshape(1)
        dq, dt = raw_disc_bwd(q, table, idx, rmax, rsum, g, ctx.sign / q.shape[0], need_dq=ctx.needs_input_grad[0],
                              need_dt=ctx.needs_input_grad[1], dt_sink=ctx.sink, lp=ctx.lp)
        return dq, dt, None, None, None


def disc_lse(q, table, idx, lp=False, sign=1.0):
    """lp=True: the bf16 compute mode (models built with compute_dtype='bf16'); default = the f32 parity mode.
    sign: the result is sign * CE (the reference returns +CE as log_qy, simple_fhvae.py:122; the intended objective -CE)."""
    return _DiscLse.apply(q, table, idx, bool(lp), float(sign))


def wgrad_bf16_(c, a, b):
    """c[M,N] (f32) += a[K,M]^T . b[K,N] for bf16 a, b whose rows are the contraction index (fhvae_wgrad_bf16)."""
    lib = load_library()
    _need_gpu(c, a, b)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and c.dtype == torch.float32
    assert a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1 and a.shape[0] == b.shape[0]
    K, M = a.shape
    N = b.shape[1]
    with _Timed("fhvae_wgrad_bf16"):
        _check(lib.fhvae_wgrad_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), M, N, K, _stream()), "fhvae_wgrad_bf16")
    return c


def wgrad_f32_(c, a, b):
    """c[M,N] (f32) += a[K,M]^T . b[K,N] for f32 a, b whose rows are the contraction index (fhvae_wgrad_f32: exact-f32 MFMA)."""
    lib = load_library()
    _need_gpu(c, a, b)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and c.dtype == torch.float32
    assert a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1 and a.shape[0] == b.shape[0]
    K, M = a.shape
    N = b.shape[1]
    with _Timed("fhvae_wgrad_f32"):
        _check(lib.fhvae_wgrad_f32(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), M, N, K, _stream()), "fhvae_wgrad_f32")
    return c


def proj_bf16(a, w, bias=None, out=None):
    """out[M,N] (f32) = a[M,K] . w[N,K]^T (+ bias) for bf16 a, w with contiguous rows (fhvae_proj_bf16: csrc/proj.hip)."""
    lib = load_library()
    _need_gpu(a, w)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1]
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    with _Timed("fhvae_proj_bf16"):
        _check(lib.fhvae_proj_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, _stream()), "fhvae_proj_bf16")
    return out


ADAM_ZERO_GRAD, ADAM_ADVANCE, ADAM_STEP_WORDS = 1, 2, 65 * 32  # FHVAE_ADAM_* of include/fhvae_hip.h
_ONES = {}


def backward(loss: torch.Tensor):
    """loss.backward() with a cached gradient seed: autograd's implicit ones_like is one fill launch per step."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _ONES.get(key)
    if one is None:
        if loss.is_cuda and torch.cuda.is_current_stream_capturing():  # (no persistent allocation inside a capture: the plain form this once)
            loss.backward()
            return
        one = _ONES[key] = torch.ones_like(loss)
    loss.backward(gradient=one)


def adam_step_(p, g, m, v, step_dev, lr, beta1, beta2, eps, grad_scale=1.0, p_lp=None, flags=0):
    """In-place fused Adam on flat f32 views (train_model.py:409-411).  step_dev: the int32 step count on the device, already
    incremented -- or, with ADAM_ADVANCE, int32[ADAM_STEP_WORDS] ([0] the count, the rest the kernel's scratch words): the launch
    counts the step itself.  ADAM_ZERO_GRAD clears g behind its use."""
    lib = load_library()
    _need_gpu(p, g, m, v, step_dev)
    if (flags & ADAM_ADVANCE) and step_dev.numel() < ADAM_STEP_WORDS:
        raise RuntimeError("ADAM_ADVANCE needs an int32[%d] step buffer (step count + scratch words)" % ADAM_STEP_WORDS)
    n = p.numel()
    if n == 0:  # an empty table shard (more ranks than rows)
        return
    with _Timed("fhvae_adam_step"):
        _check(lib.fhvae_adam_step(_p(p), _p(g), _p(m), _p(v), _p(p_lp), n, lr, beta1, beta2, eps, grad_scale, int(flags), _p(step_dev),
                                   _stream()), "fhvae_adam_step")



# ---------------------------------------------------------------------------------------------
# SURVEY 8f "next" rows: resident-pool segment sampler and closed-form mu2 estimate
# ---------------------------------------------------------------------------------------------
def segment_gather(pool, start, T, mean=None, inv_std=None, time_major=False):
    """Cut B segments of T frames out of the HBM-resident utterance pool (frames, F) at absolute frame offsets
    `start` (B,) int64, with optional fused mean/variance normalisation.  Returns (B,T,F) (and (T,B,F) if asked)."""
    lib = load_library()
    _need_gpu(pool, start, mean, inv_std)
    pool = _f32c(pool)
    B, F_ = start.shape[0], pool.shape[1]
    out = torch.empty(B, T, F_, device=pool.device, dtype=torch.float32)
    out_tm = torch.empty(T, B, F_, device=pool.device, dtype=torch.float32) if time_major else None
    with _Timed("fhvae_segment_gather"):
        _check(lib.fhvae_segment_gather(_p(pool), pool.shape[0], _p(start), _p(mean), _p(inv_std), _p(out), _p(out_tm), B, T, F_,
                                        None, _stream()), "fhvae_segment_gather")
    return (out, out_tm) if time_major else out


class Mu2Estimator:
    """Running closed-form mu2 estimate over batches (utils.py:45-60) on the device."""

    def __init__(self, num_seqs: int, dim: int, device):
        self.S, self.D = int(num_seqs), int(dim)
        self.zsum = torch.zeros(self.S, self.D, device=device, dtype=torch.float32)
        self.count = torch.zeros(self.S, device=device, dtype=torch.float32)

    def add(self, z2_mu, idx):
        lib = load_library()
        _need_gpu(z2_mu, idx)
        z2_mu = _f32c(z2_mu.detach())
        with _Timed("fhvae_mu2_accumulate"):
            _check(lib.fhvae_mu2_accumulate(_p(z2_mu), _p(idx), _p(self.zsum), _p(self.count), z2_mu.shape[0], self.S, self.D,
                                            _stream()), "fhvae_mu2_accumulate")

    def result(self, ratio: float):
        lib = load_library()
        mu2 = torch.empty_like(self.zsum)
        with _Timed("fhvae_mu2_finalize"):
            _check(lib.fhvae_mu2_finalize(_p(self.zsum), _p(self.count), _p(mu2), self.S, self.D, float(ratio), _stream()),
                   "fhvae_mu2_finalize")
        return mu2, self.count
