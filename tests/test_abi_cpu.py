"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads without a GPU and exports
every symbol include/fhvae_hip.h declares; host-side modules keep the reference's surface; the product
path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import build_ext

    build_ext.build(verbose=False)
    import hip_binding as hb

    return hb.load_library()


def _declared():
    text = open(os.path.join(ROOT, "include", "fhvae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fhvae_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    import hip_binding as hb

    names = _declared()
    assert len(names) >= 26
    for n in names:
        assert hasattr(lib, n), "libfhvae_hip.so lacks %s" % n
        assert n in hb.SIGNATURES, "hip_binding does not bind %s" % n
    assert set(hb.SIGNATURES) == set(names)
    assert lib.fhvae_abi_version() == 11
    assert lib.fhvae_strerror(-1) == b"required pointer is NULL"


def test_struct_layouts_match_header(lib):
    import hip_binding as hb

    # sizes computed from the header by hand: pointers/int64 are 8 bytes, int32 pairs packed
    assert ctypes.sizeof(hb.LstmDesc) == 8 + 5 * 8 + 3 * 8 + 4 * 4 * 8 + 8 * 8 + 8 + 7 * 8   # + hn_lp + the head_* fields (ABI 10)
    assert ctypes.sizeof(hb.LstmBwdDesc) == ctypes.sizeof(hb.LstmDesc) + 5 * 8 + 4 * 4 * 8 + 8 + 8 + 8
    assert ctypes.sizeof(hb.ElboDesc) == 5 * 8 + 3 * 8 + 4 * 8 + 5 * 8 + 2 * 8 + 5 * 8
    assert ctypes.sizeof(hb.ElboBwdDesc) == ctypes.sizeof(hb.ElboDesc) + 5 * 8 + 8 + 7 * 8 + 3 * 8


def test_argument_errors_are_reported_before_any_launch(lib):
    # NULL pointers / bad shapes return negative codes on the host: nothing touches a GPU
    assert lib.fhvae_linear_fwd(None, 1, None, 1, None, None, 1, None, 1, 1, 1, 0, 0, None) == -1
    assert lib.fhvae_mu2_gather_fwd(None, None, 0, None, 1, 1, 1, None, None) == -1
    assert lib.fhvae_disc_lse_ws_bytes(256, 4600) > 0
    assert lib.fhvae_lstm_seq_fwd(None, None) == -1
    import hip_binding as hb

    d = hb.LstmDesc()
    d.L = 9
    assert lib.fhvae_lstm_seq_fwd(ctypes.byref(d), None) == -2
    # round-3 entries: NULL operands, a contraction length the projection kernel does not take, unaligned leading dimensions,
    # a stacked-weight buffer narrower than 2D, a wgrad descriptor that is not eligible
    assert lib.fhvae_proj_bf16(None, 64, None, 64, None, None, 64, 8, 8, 64, None) == -1
    buf = (ctypes.c_float * 4096)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.fhvae_proj_bf16(p, 72, p, 72, None, p, 8, 8, 8, 72, None) == -4          # K = 72 is not a multiple of 64
    assert lib.fhvae_wgrad_f32(None, 4, None, 4, None, 4, 4, 4, 4, None) == -1
    assert lib.fhvae_wgrad_f32(p, 6, p, 8, p, 8, 4, 8, 16, None) == -4                   # lda = 6 is not a multiple of 4
    assert lib.fhvae_head_pair_weights(p, p, p, p, 8, 8, 16, None) == -2                 # ldt = 8 < 2D = 16
    assert lib.fhvae_gauss_head_bwd_pair(None, 64, None, 64, None, 64, None, 0, None, 64, None, None, None, None, 8, 64, 8, None) == -1
    assert lib.fhvae_gauss_reparam_bwd_pair(None, None, p, 8, None, None, 8, p, 16, None, None, 4, 8, None) == -1   # d_sample without eps / logvar
    assert lib.fhvae_gauss_reparam_bwd_pair(None, None, p, 4, p, p, 8, p, 16, None, None, 4, 8, None) == -2      # ABI 11: ld_s = 4 < D = 8
    assert lib.fhvae_adam_step(p, p, p, p, None, 16, 1e-3, 0.9, 0.999, 1e-8, 1.0, 4, p, None) == -2              # ABI 11: unknown flag bit
    assert lib.fhvae_adam_step(p, p, p, p, None, 16, 1e-3, 0.9, 0.999, 1e-8, 1.0, 3, None, None) == -1           # no step buffer
    w = hb.WgradDesc(None, 8, 0, None, 8, None, 8, 8, 8, 64)
    assert lib.fhvae_wgrad_desc_ok(ctypes.byref(w)) == 0
    assert lib.fhvae_disc_lse_bwd_ws_bytes(2048, 28000, 32) > 0 and lib.fhvae_disc_lse_bwd_ws_bytes(8, 8, 32) == 0
    # ABI 10: the one-pass K5 backward's workspace is capped (the whole problem up to 1.5 GiB, then query groups): c5 and a
    # rank's view of the 8-GPU shapes fit whole, 16384 queries against 10^6 rows do not
    cap = 3 << 29
    assert lib.fhvae_disc_lse_bwd_ws_bytes(2048, 1000000, 32) == (64 * 2048 * 32 + 8 * 1000000 * 33) * 4
    assert lib.fhvae_disc_lse_bwd_ws_bytes(16384, 125000, 32) <= cap
    big = lib.fhvae_disc_lse_bwd_ws_bytes(16384, 1000000, 32)
    assert 0 < big <= cap and big >= 1000000 * 33 * 4
    assert lib.fhvae_disc_lse_bwd_ws_bytes(65536, 1000000, 32) <= cap
    # a head's stacked operands in the forward's operand-cast launch: bf16 mode only, both weights, a destination
    d = hb.LstmDesc()
    d.L, d.B, d.T, d.H, d.I, d.dtype = 1, 8, 2, 8, 8, hb.F32
    for k in ("x", "hs", "cs", "gates", "pre", "head_w_mu"):
        setattr(d, k, p.value)
    for k in ("w_ih", "w_hh", "b_ih", "b_hh"):
        getattr(d, k)[0] = p.value
    assert lib.fhvae_lstm_seq_fwd(ctypes.byref(d), None) == -3
    assert lib.fhvae_elbo_colsum_rows(2048) == 1024


def test_drop_in_surface_and_no_cpu_fallback(lib):
    from fhvae import FHVAE
    from simple_fhvae import SimpleFHVAE
    from oracle.ref_cpu import FHVAERef, SimpleFHVAERef

    torch.manual_seed(3)
    m = SimpleFHVAE(32, ["16", "16"], ["16", "16"], 16, 16, ["16", "16"])  # CLI passes strings (train_model.py:146-168)
    torch.manual_seed(3)
    r = SimpleFHVAERef(32, [16, 16], [16, 16], 16, 16, [16, 16])
    assert m.model == "simple_fhvae" and m.z1_hus == [16, 16] and m.z2_dim == 16
    sd, rsd = m.state_dict(), r.state_dict()
    assert list(sd) == list(rsd)  # same keys, same order as the reference
    assert all(torch.equal(sd[k], rsd[k]) for k in sd)  # same default init under the same seed
    torch.manual_seed(4)
    f = FHVAE(4 * 6, [8, 8], [8, 8], 4, 4, [8, 8], seg_len=4, num_seqs=7)
    torch.manual_seed(4)
    fr = FHVAERef(4 * 6, [8, 8], [8, 8], 4, 4, [8, 8], seg_len=4)
    fsd = f.state_dict()
    assert f.model == "fhvae" and fsd["mu2_table"].shape == (7, 4)
    assert all(torch.equal(fsd[k], v) for k, v in fr.state_dict().items())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 4, 8), torch.tensor([0, 1]), 3, torch.tensor([1, 2]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        f(torch.zeros(2, 4, 6), torch.tensor([0, 1]), 7, 5)
