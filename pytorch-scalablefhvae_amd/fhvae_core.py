"""Host-side logic shared by the two drop-in modules (`simple_fhvae.SimpleFHVAE`, `fhvae.FHVAE`):
the persistent mu2 table, the injected-draw plumbing and the loss tail that follows the three nets
(simple_fhvae.py:86-88 and :105-124).  All arithmetic is delegated to hip_binding (HIP kernels)."""
from __future__ import annotations

import warnings
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

import hip_binding as hb

PZ1_LOGVAR = np.log(1.0 ** 2).astype(np.float32)   # simple_fhvae.py:22
PMU2_LOGVAR = np.log(1.0 ** 2).astype(np.float32)  # simple_fhvae.py:23
PZ2_LOGVAR = np.log(0.5 ** 2).astype(np.float32)   # simple_fhvae.py:88


class LocalTableOps:
    """Single-GPU mu2-table ops: gather (K4) and discriminative CE (K5) against the whole table."""

    def __init__(self, model):
        self.model = model

    def lookup(self, mu_idx, num_seqs, mu2_table=None):
        table = mu2_table if mu2_table is not None else self.model._table(num_seqs, mu_idx.device)
        return table, hb.mu2_gather(table, mu_idx)

    def disc(self, z2_mu, table, mu_idx, sign=1.0):
        return hb.disc_lse(z2_mu, table, mu_idx, lp=getattr(self.model, "compute_dtype", "f32") == "bf16", sign=sign)

    def resolve(self, z2_mu, table, mu_idx, mu2, sign=1.0):
        """(mu2 of the batch, sign * CE against the whole table): the two things the loss tail needs from the table.  One call so
        that a sharded table can serve both from ONE exchange (dist_shard.ShardedTableOps.resolve)."""
        return mu2, self.disc(z2_mu, table, mu_idx, sign)


class FHVAEBase(nn.Module):
    """Common constructor surface and loss tail.

    Positional constructor/forward signatures are the reference's (simple_fhvae.py:9-17,71-73;
    fhvae.py:5-13).  Keyword-only additions:
      num_seqs          -- create the persistent, learnable mu2 table (S, z2_dim) ~ N(0,1) at
                           construction (the reference re-draws a throw-away table on every forward,
                           simple_fhvae.py:51).  If omitted the table is created at the
                           first forward (and an optimizer built earlier will not see it).
      reference_compat  -- True (default): forward values AND gradients follow the reference
                           literally: `.detach()` on the decoder outputs and on mu2 inside the bound
                           (simple_fhvae.py:107,114) and `log_qy` = +CE (:122).  False: the intended
                           objective (decoder trained, `log_qy` = -CE).
    forward() keyword-only additions (parity injection): `mu2_table`, `eps=(eps_z2, eps_z1)`.
    """

    def _init_common(self, z1_hus, z2_hus, z1_dim, z2_dim, x_hus, num_seqs, reference_compat):
        # hus arrive as strings from the reference CLI (nargs=2 without type, train_model.py:146-168)
        self.z1_hus = [int(h) for h in z1_hus]
        self.z2_hus = [int(h) for h in z2_hus]
        self.x_hus = [int(h) for h in x_hus]
        self.z1_dim = int(z1_dim)
        self.z2_dim = int(z2_dim)
        self.pz1 = [0.0, PZ1_LOGVAR]
        self.pmu2 = [0.0, PMU2_LOGVAR]
        self.pz2 = [None, PZ2_LOGVAR]  # [mu2 of the last batch, logvar]; utils.py:58 reads pz2[1]
        self.reference_compat = bool(reference_compat)
        self._init_num_seqs = num_seqs
        self.table_ops = LocalTableOps(self)  # dist_shard.DistributedFHVAE swaps in the row-sharded ops

    def _maybe_create_table(self):
        # called at the END of __init__ so that the nets' default init consumes the RNG first
        # (same draws as the reference constructor under the same seed)
        if self._init_num_seqs is not None:
            self.mu2_table = nn.Parameter(torch.empty(int(self._init_num_seqs), self.z2_dim).normal_(0.0, 1.0))
        else:
            self.register_parameter("mu2_table", None)

    def _table(self, num_seqs: int, device) -> torch.Tensor:
        if self.mu2_table is None:
            warnings.warn(
                "mu2_table created lazily at the first forward; optimizers built before this call do not "
                "contain it (pass num_seqs= to the constructor)")
            self.mu2_table = nn.Parameter(torch.empty(int(num_seqs), self.z2_dim, device=device).normal_(0.0, 1.0))
        if self.mu2_table.shape[0] != int(num_seqs):
            raise ValueError("num_seqs=%d does not match the mu2 table (%d rows)" % (num_seqs, self.mu2_table.shape[0]))
        return self.mu2_table

    @staticmethod
    def _check_idx(mu_idx, num_seqs):
        """The reference's torch.gather raises on an out-of-range sequence index (simple_fhvae.py:53).  Indices that
        arrive on the host (the reference loop builds them there, train_model.py:445) are range-checked for free;
        device-resident indices are not (that would cost a device sync per step): out-of-range rows then read as 0."""
        t = torch.as_tensor(mu_idx)
        if not t.is_cuda and t.numel() > 0:
            lo, hi = int(t.min()), int(t.max())
            if lo < 0 or hi >= int(num_seqs):
                raise IndexError("mu_idx out of range: [%d, %d] for a mu2 table of %d rows" % (lo, hi, num_seqs))

    @staticmethod
    def _prep_inputs(x, mu_idx, num_segs):
        if not x.is_cuda:
            raise RuntimeError("FHVAE (HIP path) needs the model and inputs on a MI355X device; no CPU fallback")
        if x.dtype != torch.float32:
            raise RuntimeError("the HIP path computes in float32/bf16; got input dtype %s "
                               "(the reference's model.double() at train_model.py:438 is not supported)" % x.dtype)
        dev = x.device
        mu_idx = torch.as_tensor(mu_idx).to(device=dev, dtype=torch.int64).contiguous()
        if isinstance(num_segs, torch.Tensor):
            num_segs = num_segs.to(device=dev, dtype=torch.int64).contiguous()
        return x.contiguous(), mu_idx, num_segs

    def _draw(self, eps, B, device):
        if eps is not None:
            return eps[0].to(device), eps[1].to(device)
        # reference draw order after the table: eps_z2 then eps_z1; equal widths: one generator launch for both
        if self.z1_dim == self.z2_dim:
            e = torch.randn(2, B, self.z2_dim, device=device)
            return e[0], e[1]
        e2 = torch.randn(B, self.z2_dim, device=device)
        e1 = torch.randn(B, self.z1_dim, device=device)
        return e2, e1

    def _tail(self, x_like, layout, x_mu, x_lv, z1, z2, mu2, table, mu_idx, num_segs):
        """simple_fhvae.py:105-124 on the HIP kernels."""
        rc = self.reference_compat
        # log_qy = +CE literally (simple_fhvae.py:122) / -CE for the intended objective: the sign rides in the K5 kernels
        mu2, log_qy = self.table_ops.resolve(z2[0], table, mu_idx, mu2, 1.0 if rc else -1.0)
        lb, lpx, nk1, nk2, lpm = hb.elbo(x_like, x_mu, x_lv, z1[0], z1[1], z2[0], z2[1], mu2, num_segs, layout, rc)
        self.qz2_x = [z2[0], z2[1]]      # read by estimate_mu2_dict, utils.py:52
        self.pz2 = [mu2, PZ2_LOGVAR]     # utils.py:58
        return lb, log_qy, lpx, nk1, nk2, lpm
