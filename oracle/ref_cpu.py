"""CPU oracle for the FHVAE training hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, eager) restatement of the reference algorithm
(BurnhamG/PyTorch-ScalableFHVAE).  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``pytorch-scalablefhvae_amd/``) never imports anything from
``oracle/`` and has no CPU fallback.

Pinning status
--------------
* Loss arithmetic + fully-connected ``SimpleFHVAE``: PINNED.  ``tests/golden/*.npz``
  were produced by importing the reference's ``simple_fhvae.SimpleFHVAE`` in the build
  container (``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks this
  restatement against them (forward 6-tuple, loss, and gradients).
* LSTM ``FHVAE``: PARITY UNPINNED BY THE REFERENCE.  ``fhvae.py:4-14`` is a stub that
  raises ``NotImplementedError``; there is nothing to import or to take vectors from.
  The architecture below (``FHVAERef``) is *defined by this build* on ``torch.nn.LSTM``
  CPU semantics (gate order i,f,g,o; ``b_ih + b_hh``) and shares the pinned loss
  arithmetic with ``SimpleFHVAERef``.

Every function cites the reference file:line it restates (paths relative to the
reference repository root).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

# ---------------------------------------------------------------------------------------------
# constants (simple_fhvae.py:22-23, :88)
# ---------------------------------------------------------------------------------------------
#: prior log-variance of z1 and of mu2: ``np.log(1.0 ** 2).astype(np.float32)``  (simple_fhvae.py:22-23)
PZ1_LOGVAR = np.log(1.0 ** 2).astype(np.float32)
PMU2_LOGVAR = np.log(1.0 ** 2).astype(np.float32)
#: prior log-variance of z2 around mu2: ``np.log(0.5 ** 2).astype(np.float32)`` (simple_fhvae.py:88)
PZ2_LOGVAR = np.log(0.5 ** 2).astype(np.float32)
LOG_2PI = np.log(2 * np.pi)  # python float64, as in simple_fhvae.py:59


def log_gauss(x: torch.Tensor, mu=0.0, logvar=0.0) -> torch.Tensor:
    """``log N(x; mu, exp(logvar))`` -- simple_fhvae.py:56-60.

    The reference evaluates ``np.exp(logvar)`` -- numpy's exp on a detached CPU tensor (or on a
    numpy scalar).  numpy's float32 ``exp`` and ``torch.exp`` differ by 1 ulp on ~39 % of inputs
    (measured), so a detached tensor takes the numpy route here too (bit-exact against the golden
    vectors); a tensor that carries gradient (``reference_detach=False``) must use ``torch.exp``.
    """
    if isinstance(logvar, torch.Tensor) and not logvar.requires_grad:
        var = torch.from_numpy(np.exp(logvar.numpy()))
    elif isinstance(logvar, torch.Tensor):
        var = torch.exp(logvar)
    else:
        var = np.exp(logvar)
    return -0.5 * (LOG_2PI + logvar + torch.pow(x - mu, 2) / var)


def kld(p_mu, p_logvar, q_mu, q_logvar) -> torch.Tensor:
    """``D_KL(p || q)`` of two diagonal Gaussians -- simple_fhvae.py:62-69 (``q_logvar`` is a
    numpy float32 scalar at both call sites)."""
    return -0.5 * (
        1
        + p_logvar
        - q_logvar
        - (torch.pow(p_mu - q_mu, 2) + torch.exp(p_logvar)) / np.exp(q_logvar)
    )


def mu2_gather(table: torch.Tensor, mu_idx: torch.Tensor) -> torch.Tensor:
    """``torch.gather(table, 0, stack([idx]*16, 1))`` == ``table[idx]`` -- simple_fhvae.py:53
    (the reference hard-codes 16 columns; the restatement uses the table's width)."""
    return torch.gather(table, 0, torch.stack([mu_idx] * table.shape[1], 1))


def elbo_terms(
    x, x_mu, x_logvar, z1_mu, z1_logvar, z2_mu, z2_logvar, mu2, num_segs, reference_detach=True
):
    """Variational lower bound block -- simple_fhvae.py:105-116.

    Returns ``(lower_bound, log_px_z, neg_kld_z1, neg_kld_z2, log_pmu2)``, each ``(B,)``.
    ``reference_detach=True`` reproduces the reference's ``.detach()`` on ``mu2`` inside
    ``log_pmu2`` (:107) and on ``x_mu, x_logvar`` inside ``log_px_z`` (:114); ``False`` is the
    objective with the gradients left attached (values identical).
    """
    mu2_p = mu2.detach() if reference_detach else mu2
    log_pmu2 = torch.sum(log_gauss(mu2_p, 0.0, PMU2_LOGVAR), dim=1)
    neg_kld_z2 = -1 * torch.sum(kld(z2_mu, z2_logvar, mu2, PZ2_LOGVAR), dim=1)
    neg_kld_z1 = -1 * torch.sum(kld(z1_mu, z1_logvar, 0.0, PZ1_LOGVAR), dim=1)
    if reference_detach:
        log_px_z = torch.sum(log_gauss(x, x_mu.detach(), x_logvar.detach()), dim=(1, 2))
    else:
        log_px_z = torch.sum(log_gauss(x, x_mu, x_logvar), dim=(1, 2))
    lower_bound = log_px_z + neg_kld_z1 + neg_kld_z2 + log_pmu2 / num_segs
    return lower_bound, log_px_z, neg_kld_z1, neg_kld_z2, log_pmu2


def disc_logits(z2_mu: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """``logits[b,s] = -sum_d (z2_mu[b,d]-table[s,d])^2 / (2*exp(pz2_logvar))`` --
    simple_fhvae.py:119-121, materialising the ``(B,S,D)`` tensor exactly like the reference."""
    logits = torch.unsqueeze(z2_mu, 1) - torch.unsqueeze(table, 0)
    logits = -1 * torch.pow(logits, 2) / (2 * np.exp(PZ2_LOGVAR))
    return torch.sum(logits, dim=-1)


def disc_loss(z2_mu: torch.Tensor, table: torch.Tensor, mu_idx: torch.Tensor) -> torch.Tensor:
    """``log_qy = CrossEntropyLoss(mean)(logits, mu_idx)`` -- simple_fhvae.py:37,:122 (a scalar)."""
    return nn.functional.cross_entropy(disc_logits(z2_mu, table), mu_idx)


def disc_loss_chunked(z2_mu, table, mu_idx, chunk: int = 4096) -> torch.Tensor:
    """Same value as :func:`disc_loss` without the ``(B,S,D)`` temporary (for table sizes whose
    temporary does not fit in host RAM).  Online log-sum-exp over S chunks."""
    B = z2_mu.shape[0]
    m = torch.full((B,), -float("inf"), dtype=z2_mu.dtype)
    s = torch.zeros((B,), dtype=z2_mu.dtype)
    for s0 in range(0, table.shape[0], chunk):
        lg = disc_logits(z2_mu, table[s0 : s0 + chunk])
        m_new = torch.maximum(m, lg.max(dim=1).values)
        s = s * torch.exp(m - m_new) + torch.exp(lg - m_new[:, None]).sum(dim=1)
        m = m_new
    lse = m + torch.log(s)
    tgt = -1 * torch.pow(z2_mu - table[mu_idx], 2).sum(dim=1) / (2 * np.exp(PZ2_LOGVAR))
    return (lse - tgt).mean()


def loss_function(lower_bound, log_qy, alpha=10.0):
    """``-mean(lower_bound + alpha*log_qy)`` -- train_model.py:243-251."""
    return -1 * torch.mean(lower_bound + alpha * log_qy)


def check_terminate(epoch, best_epoch, patience, epochs) -> bool:
    """train_model.py:254-261."""
    if (epoch - 1) - best_epoch > patience:
        return True
    if epoch > epochs:
        return True
    return False


def check_best(val_lower_bound, best_val_lb) -> bool:
    """utils.py:14-17."""
    return bool(torch.mean(val_lower_bound) > best_val_lb)


def estimate_mu2(z2_mu: torch.Tensor, mu_idx: torch.Tensor, num_seqs: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Closed-form mu2 posterior mean per sequence -- utils.py:45-60:
    ``mu2[y] = sum_{segments of y} z2_mu / (n_y + exp(pz2_logvar)/exp(pmu2_logvar))``.
    Returns ``(mu2 (S,D), nseg (S,))``; sequences with no segment get 0."""
    S, D = num_seqs, z2_mu.shape[1]
    zsum = torch.zeros(S, D, dtype=z2_mu.dtype).index_add_(0, mu_idx, z2_mu)
    n = torch.zeros(S, dtype=z2_mu.dtype).index_add_(0, mu_idx, torch.ones_like(z2_mu[:, 0]))
    r = float(np.exp(PZ2_LOGVAR) / np.exp(PMU2_LOGVAR))
    mu2 = torch.where(n[:, None] > 0, zsum / (n[:, None] + r), torch.zeros_like(zsum))
    return mu2, n


# ---------------------------------------------------------------------------------------------
# shared forward tail: everything after the three nets (simple_fhvae.py:105-124)
# ---------------------------------------------------------------------------------------------
def _forward_tail(x, mu_idx, num_segs, table, mu2, z1, z2, xd, reference_compat):
    z1_mu, z1_logvar = z1
    z2_mu, z2_logvar = z2
    x_mu, x_logvar = xd
    lower_bound, log_px_z, neg_kld_z1, neg_kld_z2, log_pmu2 = elbo_terms(
        x, x_mu, x_logvar, z1_mu, z1_logvar, z2_mu, z2_logvar, mu2, num_segs, reference_detach=reference_compat
    )
    ce = disc_loss(z2_mu, table, mu_idx)
    # reference: log_qy is the *positive* CE mean (simple_fhvae.py:122); the intended objective
    # (reference_compat=False) is log q(y|z2) = -CE
    log_qy = ce if reference_compat else -ce
    return lower_bound, log_qy, log_px_z, neg_kld_z1, neg_kld_z2, log_pmu2


def gauss_sample(mu, logvar, eps):
    """``mu + eps*exp(0.5*logvar)`` -- simple_fhvae.py:214-216 (``eps`` injected instead of drawn)."""
    return mu + eps * torch.exp(0.5 * logvar)


# ---------------------------------------------------------------------------------------------
# fully-connected model (restates simple_fhvae.py:8-244); state-dict keys identical
# ---------------------------------------------------------------------------------------------
class _FC(nn.Module):  # simple_fhvae.py:127-134
    def __init__(self, i, o):
        super().__init__()
        self.linear = nn.Linear(i, o)

    def forward(self, x):
        return torch.relu(self.linear(x))


class _PreNet(nn.Module):  # simple_fhvae.py:137-190, :219-244 (fc1, fc2)
    def __init__(self, i, hus):
        super().__init__()
        self.fc1 = _FC(i, hus[0])
        self.fc2 = _FC(hus[0], hus[1])

    def forward(self, x):
        return self.fc2(self.fc1(x))


class _Gauss(nn.Module):  # simple_fhvae.py:193-216
    def __init__(self, i, d):
        super().__init__()
        self.mulayer = nn.Linear(i, d)
        self.logvar_layer = nn.Linear(i, d)

    def forward(self, h):
        return self.mulayer(h), self.logvar_layer(h)


class SimpleFHVAERef(nn.Module):
    """Restatement of ``SimpleFHVAE`` (simple_fhvae.py:8-124) with the random draws injected.

    Differences from the reference, none of which change values at ``z1_dim == z2_dim == 16``:
    the z1 pre-encoder is sized ``input_size + z2_dim`` (reference :31 uses ``z1_dim`` but feeds
    ``z2_sample``), and the gather width follows the table (reference :53 hard-codes 16).
    Sub-module construction order (and hence default-init RNG order) matches :31-36.
    """

    def __init__(self, input_size, z1_hus=(128, 128), z2_hus=(128, 128), z1_dim=16, z2_dim=16, x_hus=(128, 128)):
        super().__init__()
        self.model = "simple_fhvae"
        self.z1_hus, self.z2_hus, self.x_hus = list(z1_hus), list(z2_hus), list(x_hus)
        self.z1_dim, self.z2_dim = z1_dim, z2_dim
        self.z1_pre_encoder = _PreNet(input_size + z2_dim, self.z1_hus)
        self.z2_pre_encoder = _PreNet(input_size, self.z2_hus)
        self.z1_gauss_layer = _Gauss(self.z1_hus[1], z1_dim)
        self.z2_gauss_layer = _Gauss(self.z2_hus[1], z2_dim)
        self.pre_decoder = _PreNet(z1_dim + z2_dim, self.x_hus)
        self.dec_gauss_layer = _Gauss(self.x_hus[1], input_size)

    def forward(self, x, mu_idx, num_seqs, num_segs, *, mu2_table, eps_z2, eps_z1, reference_compat=True):
        B = x.shape[0]
        xf = x.reshape(B, -1)
        mu2 = mu2_gather(mu2_table, mu_idx)
        z2_mu, z2_logvar = self.z2_gauss_layer(self.z2_pre_encoder(xf))
        z2_sample = gauss_sample(z2_mu, z2_logvar, eps_z2)
        z1_mu, z1_logvar = self.z1_gauss_layer(self.z1_pre_encoder(torch.cat([xf, z2_sample], -1)))
        z1_sample = gauss_sample(z1_mu, z1_logvar, eps_z1)
        x_mu, x_logvar = self.dec_gauss_layer(self.pre_decoder(torch.cat([z1_sample, z2_sample], -1)))
        x_mu = x_mu.view(-1, x.shape[1], x.shape[2])
        x_logvar = x_logvar.view(-1, x.shape[1], x.shape[2])
        self.qz2_x = [z2_mu, z2_logvar]  # utils.py:52 reads these after a forward
        self.pz2 = [mu2, PZ2_LOGVAR]
        self.pmu2 = [0.0, PMU2_LOGVAR]
        return _forward_tail(
            x, mu_idx, num_segs, mu2_table, mu2, (z1_mu, z1_logvar), (z2_mu, z2_logvar), (x_mu, x_logvar), reference_compat
        )


# ---------------------------------------------------------------------------------------------
# LSTM model -- NO reference body exists (fhvae.py:14 raises).  Defined by this build.
# ---------------------------------------------------------------------------------------------
class _LSTMNet(nn.Module):
    def __init__(self, i, hus):
        super().__init__()
        assert len(set(hus)) == 1, "all LSTM layers of a net share one hidden size"
        self.lstm = nn.LSTM(i, hus[0], num_layers=len(hus), batch_first=True)


class FHVAERef(nn.Module):
    """LSTM FHVAE with the constructor signature of the stub ``fhvae.py:5-13``.

    Wiring (docstrings of the FC stand-ins, simple_fhvae.py:175,229, say "Concatenation of hidden
    states of all layers"; the sequential structure is the design of the papers cited at
    README.md:9-10):
      z2 encoder : LSTM over x (B,T,F)            -> cat_l h_T^l (B, L*H) -> Gaussian head
      z1 encoder : LSTM over [x_t || z2_sample]   -> cat_l h_T^l          -> Gaussian head
      decoder    : LSTM over [z1 || z2] tiled T x -> top-layer h_t per frame -> per-frame Gaussian head (F)
    ``input_size = T*F`` (train_model.py:398); ``seg_len`` (keyword-only, default 20,
    train_model.py:120) fixes ``F = input_size // seg_len``.
    """

    def __init__(self, input_size, z1_hus, z2_hus, z1_dim, z2_dim, x_hus, *, seg_len=20):
        super().__init__()
        self.model = "fhvae"
        self.z1_hus, self.z2_hus, self.x_hus = [int(h) for h in z1_hus], [int(h) for h in z2_hus], [int(h) for h in x_hus]
        self.z1_dim, self.z2_dim = int(z1_dim), int(z2_dim)
        self.seg_len = seg_len
        assert input_size % seg_len == 0
        self.n_feat = F = int(input_size) // seg_len
        self.z1_pre_encoder = _LSTMNet(F + self.z2_dim, self.z1_hus)
        self.z2_pre_encoder = _LSTMNet(F, self.z2_hus)
        self.z1_gauss_layer = _Gauss(sum(self.z1_hus), self.z1_dim)
        self.z2_gauss_layer = _Gauss(sum(self.z2_hus), self.z2_dim)
        self.pre_decoder = _LSTMNet(self.z1_dim + self.z2_dim, self.x_hus)
        self.dec_gauss_layer = _Gauss(self.x_hus[-1], F)

    @staticmethod
    def _final_h(h_n):
        return torch.cat([h_n[l] for l in range(h_n.shape[0])], dim=-1)

    def forward(self, x, mu_idx, num_seqs, num_segs, *, mu2_table, eps_z2, eps_z1, reference_compat=True):
        B, T, F = x.shape
        mu2 = mu2_gather(mu2_table, mu_idx)
        _, (h_n, _) = self.z2_pre_encoder.lstm(x)
        z2_mu, z2_logvar = self.z2_gauss_layer(self._final_h(h_n))
        z2_sample = gauss_sample(z2_mu, z2_logvar, eps_z2)
        z1_in = torch.cat([x, z2_sample[:, None, :].expand(B, T, -1)], dim=-1)
        _, (h_n, _) = self.z1_pre_encoder.lstm(z1_in)
        z1_mu, z1_logvar = self.z1_gauss_layer(self._final_h(h_n))
        z1_sample = gauss_sample(z1_mu, z1_logvar, eps_z1)
        dec_in = torch.cat([z1_sample, z2_sample], dim=-1)[:, None, :].expand(B, T, -1)
        out, _ = self.pre_decoder.lstm(dec_in)
        x_mu, x_logvar = self.dec_gauss_layer(out)
        self.qz2_x = [z2_mu, z2_logvar]
        self.pz2 = [mu2, PZ2_LOGVAR]
        self.pmu2 = [0.0, PMU2_LOGVAR]
        return _forward_tail(
            x, mu_idx, num_segs, mu2_table, mu2, (z1_mu, z1_logvar), (z2_mu, z2_logvar), (x_mu, x_logvar), reference_compat
        )


# ---------------------------------------------------------------------------------------------
# deterministic closed-form tensors (regenerable anywhere without torch RNG) for fixtures/tests
# ---------------------------------------------------------------------------------------------
def det_tensor(shape: Sequence[int], seed: float, scale: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    """``scale * sin(seed + 0.37*i + 0.011*i^2 mod 1000)`` over the flat index, computed in float64
    then cast: identical on every machine."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.sin(seed + 0.37 * i + np.mod(0.011 * i * i, 1000.0)) * scale
    return torch.from_numpy(v.reshape(tuple(shape))).to(dtype)


def det_index(n: int, high: int, seed: int) -> torch.Tensor:
    i = np.arange(n, dtype=np.int64)
    return torch.from_numpy((i * 7919 + seed * 104729 + (i * i) % 31) % high)


def fill_state_dict_det(model: nn.Module, seed: float = 1.0) -> None:
    """Overwrite every parameter with :func:`det_tensor` values scaled like default init."""
    with torch.no_grad():
        for k, (name, p) in enumerate(sorted(model.named_parameters())):
            fan = p.shape[-1] if p.dim() > 1 else p.shape[0]
            p.copy_(det_tensor(p.shape, seed + 13.0 * k, 1.0 / math.sqrt(max(fan, 1)), p.dtype))


# ---------------------------------------------------------------------------------------------
# one full training step on CPU (used by bench.py's cpu_baseline leg and by trajectory tests)
# ---------------------------------------------------------------------------------------------
TERM_NAMES = ("lower_bound", "log_qy", "log_px_z", "neg_kld_z1", "neg_kld_z2", "log_pmu2")  # fhvae.py:185 return order


def train_step(model, opt, table, x, mu_idx, num_segs, eps_z2, eps_z1, alpha=10.0, reference_compat=False, terms=None):
    """zero_grad -> forward -> loss_function -> backward -> Adam step (train_model.py:446-454).  `terms`: a dict that
    receives the batch mean of every term the forward returned (TERM_NAMES)."""
    opt.zero_grad(set_to_none=True)
    out = model(x, mu_idx, table.shape[0], num_segs, mu2_table=table, eps_z2=eps_z2, eps_z1=eps_z1,
                reference_compat=reference_compat)
    loss = loss_function(out[0], out[1], alpha)
    loss.backward()
    opt.step()
    if terms is not None:
        terms.update({n: o.detach().mean().item() for n, o in zip(TERM_NAMES, out)})
    return loss.detach(), out[0].detach()
