"""Drop-in `simple_fhvae.SimpleFHVAE` on the MI355X HIP kernels.

Same module name, class names, constructor/forward signatures, attribute names and state-dict keys
as the reference's simple_fhvae.py:8-244; the arithmetic runs in libfhvae_hip.so (no ATen compute).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

import hip_binding as hb
from fhvae_core import FHVAEBase


class VariableLinearLayer(nn.Module):
    """Linear + ReLU (simple_fhvae.py:127-134); `self.linear` only holds the parameters."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.linear = nn.Linear(in_dim, out_dim)

    def forward(self, x):
        return hb.linear(x, self.linear.weight, self.linear.bias, relu=True)


class LatentSegPreEncoder(nn.Module):
    """z1 pre-encoder (simple_fhvae.py:137-164)."""

    def __init__(self, input_size: int, hus: List[int] = None):
        super().__init__()
        self.hus = [1024, 1024] if hus is None else hus
        self.fc1 = VariableLinearLayer(input_size, self.hus[0])
        self.fc2 = VariableLinearLayer(self.hus[0], self.hus[1])

    def forward(self, x: torch.Tensor, lat_seq: torch.Tensor):
        out = torch.cat([x.reshape(-1, x.shape[1] * x.shape[2]), lat_seq], dim=-1)
        return self.fc2(self.fc1(out))


class LatentSeqPreEncoder(nn.Module):
    """z2 pre-encoder (simple_fhvae.py:167-190)."""

    def __init__(self, input_size, hus: List[int] = None):
        super().__init__()
        hus = [1024, 1024] if hus is None else hus
        self.fc1 = VariableLinearLayer(input_size, hus[0])
        self.fc2 = VariableLinearLayer(hus[0], hus[1])

    def forward(self, x):
        return self.fc2(self.fc1(x.reshape(-1, x.shape[1] * x.shape[2])))


class GaussianLayer(nn.Module):
    """mu / logvar heads + reparameterised sample (simple_fhvae.py:193-216).  `eps` may be injected;
    `sample=False` skips the draw (the decoder's x_sample is never used, simple_fhvae.py:102)."""

    def __init__(self, input_size: int, dim: int):
        super().__init__()
        self.mulayer = nn.Linear(input_size, dim)
        self.logvar_layer = nn.Linear(input_size, dim)

    def forward(self, input_layer: torch.Tensor, eps=None, sample=True, input_lp=None, shadows=None):
        """input_lp: optional bf16 copy of input_layer (compute_dtype='bf16' models): bf16 MFMA operands; shadows: the stacked
        bf16 copies of this layer's weights if the net in front of it already made them this step (hip_binding.lstm_seq)."""
        if sample and eps is None:
            eps = torch.randn(input_layer.shape[0], self.mulayer.out_features, device=input_layer.device)
        return hb.gauss_head(input_layer, self.mulayer.weight, self.mulayer.bias, self.logvar_layer.weight,
                             self.logvar_layer.bias, eps if sample else None, h_lp=input_lp, shadows=shadows)

    def head_weights(self):
        return self.mulayer.weight, self.logvar_layer.weight


class PreDecoder(nn.Module):
    """Pre-stochastic decoder (simple_fhvae.py:219-244)."""

    def __init__(self, input_size: int, hus: List[int] = None):
        super().__init__()
        hus = [1024, 1024] if hus is None else hus
        self.fc1 = VariableLinearLayer(input_size, hus[0])
        self.fc2 = VariableLinearLayer(hus[0], hus[1])

    def forward(self, lat_seg: torch.Tensor, lat_seq: torch.Tensor):
        return self.fc2(self.fc1(torch.cat([lat_seg, lat_seq], -1)))


class SimpleFHVAE(FHVAEBase):
    def __init__(self, input_size, z1_hus=[128, 128], z2_hus=[128, 128], z1_dim=16, z2_dim=16, x_hus=[128, 128], *,
                 num_seqs=None, reference_compat=True):
        super().__init__()
        self.model = "simple_fhvae"
        self._init_common(z1_hus, z2_hus, z1_dim, z2_dim, x_hus, num_seqs, reference_compat)
        input_size = int(input_size)
        # construction order = simple_fhvae.py:31-36 (same default-init draws under the same seed);
        # z1 pre-encoder sized with z2_dim, which is what it is actually fed (:94; reference :31 uses z1_dim)
        self.z1_pre_encoder = LatentSegPreEncoder(input_size + self.z2_dim, self.z1_hus)
        self.z2_pre_encoder = LatentSeqPreEncoder(input_size, self.z2_hus)
        self.z1_gauss_layer = GaussianLayer(self.z1_hus[1], self.z1_dim)
        self.z2_gauss_layer = GaussianLayer(self.z2_hus[1], self.z2_dim)
        self.pre_decoder = PreDecoder(self.z1_dim + self.z2_dim, self.x_hus)
        self.dec_gauss_layer = GaussianLayer(self.x_hus[1], input_size)
        self._maybe_create_table()

    def mu2_lookup(self, mu_idx: torch.Tensor, z2_dim: int, num_seqs: int, init_std: float = 1.0, mu2_table=None):
        """Table + gathered rows (simple_fhvae.py:39-54); the table persists instead of being redrawn."""
        return self.table_ops.lookup(mu_idx, num_seqs, mu2_table)

    def encode(self, x: torch.Tensor):
        """Inference-only latent extraction (eval_model.py:57-59 TODOs; used by utils.estimate_mu2_dict, utils.py:51-52):
        returns (z1_mu, z2_mu) with z1 conditioned on the posterior MEAN of z2."""
        x, _, _ = self._prep_inputs(x, torch.zeros(x.shape[0], dtype=torch.int64), 1)
        z2_mu, z2_logvar, _ = self.z2_gauss_layer(self.z2_pre_encoder(x), sample=False)
        z1_mu, z1_logvar, _ = self.z1_gauss_layer(self.z1_pre_encoder(x, z2_mu), sample=False)
        self.qz2_x = [z2_mu, z2_logvar]
        return z1_mu, z2_mu

    def forward(self, x: torch.Tensor, mu_idx: torch.Tensor, num_seqs: int, num_segs, *, mu2_table=None, eps=None):
        self._check_idx(mu_idx, num_seqs)
        x, mu_idx, num_segs = self._prep_inputs(x, mu_idx, num_segs)
        B, T, F_ = x.shape
        mu2_table, mu2 = self.mu2_lookup(mu_idx, self.z2_dim, num_seqs, mu2_table=mu2_table)
        e2, e1 = self._draw(eps, B, x.device)

        z2_pre_out = self.z2_pre_encoder(x)
        z2_mu, z2_logvar, z2_sample = self.z2_gauss_layer(z2_pre_out, e2)
        z1_pre_out = self.z1_pre_encoder(x, z2_sample)
        z1_mu, z1_logvar, z1_sample = self.z1_gauss_layer(z1_pre_out, e1)
        x_pre_out = self.pre_decoder(z1_sample, z2_sample)
        x_mu, x_logvar, _ = self.dec_gauss_layer(x_pre_out, sample=False)  # (B, T*F) == (B,T,F) batch-major

        layout = (B, T, F_, (T * F_, F_), (T * F_, F_))
        return self._tail(x, layout, x_mu, x_logvar, (z1_mu, z1_logvar), (z2_mu, z2_logvar), mu2, mu2_table, mu_idx,
                          num_segs)
