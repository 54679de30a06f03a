"""Drop-in `fhvae.FHVAE`: the LSTM Scalable-FHVAE on the MI355X HIP kernels.

The reference class is a stub (fhvae.py:4-14: constructor signature, then NotImplementedError); the
train loop constructs and calls it exactly like SimpleFHVAE (train_model.py:400-407, :447-449), so
the constructor and forward signatures and the 6-tuple returned are the contract.  The body is
defined by this build (DESIGN.md "FHVAE architecture"; CPU restatement: oracle/ref_cpu.py FHVAERef):

  z2 encoder : L-layer LSTM over x (B,T,F)          -> concat_l h^l_T -> Gaussian head (z2_dim)
  z1 encoder : L-layer LSTM over [x_t || z2_sample] -> concat_l h^l_T -> Gaussian head (z1_dim)
  decoder    : L-layer LSTM over [z1 || z2] (same input every step) -> top-layer h_t -> per-frame
               Gaussian head (F)
Parameters live in `*.lstm.{weight_ih_l{k},weight_hh_l{k},bias_ih_l{k},bias_hh_l{k}}` (torch.nn.LSTM
names and default init, so a state dict moves between this module and the CPU oracle unchanged).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

import hip_binding as hb
from fhvae_core import FHVAEBase
from simple_fhvae import GaussianLayer


class LSTMParams(nn.Module):
    """Parameter container with torch.nn.LSTM's names, shapes, registration order and default init
    (uniform(-1/sqrt(H), 1/sqrt(H)) over parameters in registration order)."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        k = 1.0 / math.sqrt(hidden_size)
        for l in range(num_layers):
            i = input_size if l == 0 else hidden_size
            for name, shape in (("weight_ih", (4 * hidden_size, i)), ("weight_hh", (4 * hidden_size, hidden_size)),
                                ("bias_ih", (4 * hidden_size,)), ("bias_hh", (4 * hidden_size,))):
                self.register_parameter("%s_l%d" % (name, l), nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def flat(self):
        out = []
        for l in range(self.num_layers):
            out += [getattr(self, "%s_l%d" % (n, l)) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return out


class LSTMNet(nn.Module):
    def __init__(self, input_size: int, hus):
        super().__init__()
        if len(set(hus)) != 1:
            raise ValueError("all layers of one LSTM net must share the hidden size (got %s)" % (hus,))
        self.lstm = LSTMParams(input_size, hus[0], len(hus))

    def forward(self, x_tm, xc, T, dtype=hb.F32, top=2, head=None):
        """x_tm (T,B,I) time-major or None; xc (B,Ic) constant-over-time extra input or None.
        Returns (top-layer h_t (T,B,H), concat of final h of all layers (B, L*H)).  `top`, `head`: hip_binding.lstm_seq."""
        return hb.lstm_seq(x_tm, xc, T, self.lstm.flat(), dtype, top, head.head_weights() if (head is not None and dtype == hb.BF16) else None)


class FHVAE(FHVAEBase):
    def __init__(self, input_size: int, z1_hus: list, z2_hus: list, z1_dim: int, z2_dim: int, x_hus: list, *,
                 seg_len: int = 20, num_seqs=None, reference_compat=True, compute_dtype: str = "f32"):
        super().__init__()
        self.model = "fhvae"
        self._init_common(z1_hus, z2_hus, z1_dim, z2_dim, x_hus, num_seqs, reference_compat)
        input_size = int(input_size)
        if input_size % seg_len:
            raise ValueError("input_size=%d is not a multiple of seg_len=%d" % (input_size, seg_len))
        if compute_dtype not in ("f32", "bf16"):
            raise ValueError("compute_dtype must be 'f32' (exact-f32 MFMA, parity mode) or 'bf16'")
        self.compute_dtype = compute_dtype  # MFMA operand type of the LSTM nets (master weights stay f32)
        self.seg_len = int(seg_len)         # train_model.py:120 (--seg-len, default 20)
        self.n_feat = input_size // seg_len  # input_size = T*F, train_model.py:398
        F_ = self.n_feat
        # same construction order as SimpleFHVAE (simple_fhvae.py:31-36)
        self.z1_pre_encoder = LSTMNet(F_ + self.z2_dim, self.z1_hus)
        self.z2_pre_encoder = LSTMNet(F_, self.z2_hus)
        self.z1_gauss_layer = GaussianLayer(sum(self.z1_hus), self.z1_dim)
        self.z2_gauss_layer = GaussianLayer(sum(self.z2_hus), self.z2_dim)
        self.pre_decoder = LSTMNet(self.z1_dim + self.z2_dim, self.x_hus)
        self.dec_gauss_layer = GaussianLayer(self.x_hus[-1], F_)
        self._maybe_create_table()

    def mu2_lookup(self, mu_idx: torch.Tensor, z2_dim: int, num_seqs: int, init_std: float = 1.0, mu2_table=None):
        """Table + gathered rows (simple_fhvae.py:39-54); the table persists instead of being redrawn."""
        return self.table_ops.lookup(mu_idx, num_seqs, mu2_table)

    def encode(self, x: torch.Tensor):
        """Inference-only latent extraction (eval_model.py:57-59 TODOs; used by utils.estimate_mu2_dict, utils.py:51-52):
        returns (z1_mu, z2_mu) with z1 conditioned on the posterior MEAN of z2."""
        x, _, _ = self._prep_inputs(x, torch.zeros(x.shape[0], dtype=torch.int64), 1)
        T = x.shape[1]
        dt = hb.BF16 if self.compute_dtype == "bf16" else hb.F32
        x_tm = hb.to_time_major(x, with_bf16=dt == hb.BF16)
        _, hn2 = self.z2_pre_encoder(x_tm, None, T, dt)
        z2_mu, z2_logvar, _ = self.z2_gauss_layer(hn2, sample=False)
        _, hn1 = self.z1_pre_encoder(x_tm, z2_mu, T, dt)
        z1_mu, z1_logvar, _ = self.z1_gauss_layer(hn1, sample=False)
        self.qz2_x = [z2_mu, z2_logvar]
        return z1_mu, z2_mu

    def forward(self, x: torch.Tensor, mu_idx: torch.Tensor, num_seqs: int, num_segs, *, mu2_table=None, eps=None):
        self._check_idx(mu_idx, num_seqs)
        x, mu_idx, num_segs = self._prep_inputs(x, mu_idx, num_segs)
        B, T, F_ = x.shape
        if F_ != self.n_feat:
            raise ValueError("x has %d features per frame, model was built for %d" % (F_, self.n_feat))
        mu2_table, mu2 = self.mu2_lookup(mu_idx, self.z2_dim, num_seqs, mu2_table=mu2_table)
        e2, e1 = self._draw(eps, B, x.device)

        dt = hb.BF16 if self.compute_dtype == "bf16" else hb.F32
        x_tm = hb.to_time_major(x, with_bf16=dt == hb.BF16)  # (T,B,F): contiguous per-step tiles for the step-fused cells
        # the encoders only use their final states; in bf16 mode the decoder's per-frame head reads the bf16 states, so the
        # f32 copy of the per-step states is not written at all (top=0 / top=1)
        # bf16 mode: the latent heads contract bf16 copies of the final states too (the nets that produced them ran on bf16
        # operands; hip_binding.gauss_head's condition on the sizes)
        # (the nets' forward leaves that copy beside hn: `_fh_lp`)
        lp = (lambda h, dim: (getattr(h, "_fh_lp", None) if getattr(h, "_fh_lp", None) is not None else hb.cast_bf16(h))
              if (dt == hb.BF16 and h.shape[1] % 8 == 0 and dim % 8 == 0) else None)
        # (each net's operand-cast launch also makes the stacked bf16 weights of the head behind it: `_fh_head`)
        _, hn2 = self.z2_pre_encoder(x_tm, None, T, dt, top=0, head=self.z2_gauss_layer)
        z2_mu, z2_logvar, z2_sample = self.z2_gauss_layer(hn2, e2, input_lp=lp(hn2, self.z2_dim), shadows=getattr(hn2, "_fh_head", None))
        _, hn1 = self.z1_pre_encoder(x_tm, z2_sample, T, dt, top=0, head=self.z1_gauss_layer)
        z1_mu, z1_logvar, z1_sample = self.z1_gauss_layer(hn1, e1, input_lp=lp(hn1, self.z1_dim), shadows=getattr(hn1, "_fh_head", None))
        lp_head = dt == hb.BF16 and self.x_hus[-1] % 8 == 0 and F_ % 8 == 0  # (hip_binding.gauss_head's condition)
        hs_top, _ = self.pre_decoder(None, torch.cat([z1_sample, z2_sample], dim=-1), T, dt, top=1 if lp_head else 2,
                                     head=self.dec_gauss_layer)
        H = hs_top.shape[-1]
        hs_lp = getattr(hs_top, "_fh_lp", None)  # bf16 mode: the top layer's h in bf16 = the per-frame head's operand
        x_mu, x_logvar, _ = self.dec_gauss_layer(hs_top.reshape(T * B, H), sample=False,  # (T*B, F) time-major
                                                 input_lp=hs_lp.reshape(T * B, H) if hs_lp is not None else None,
                                                 shadows=getattr(hs_top, "_fh_head", None))

        layout = (B, T, F_, (F_, B * F_), (F_, B * F_))  # x_tm and x_mu/x_logvar are all time-major
        return self._tail(x_tm, layout, x_mu, x_logvar, (z1_mu, z1_logvar), (z2_mu, z2_logvar), mu2, mu2_table, mu_idx,
                          num_segs)
