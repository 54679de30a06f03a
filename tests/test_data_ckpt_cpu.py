"""CPU tests of the SURVEY 8f host-side rows: scp/npy dataset format + segmenting rules (datasets.py of the reference)
and the checkpoint layout (utils.py:63-152).  The reference's datasets/utils modules cannot be imported here
(torchaudio / kaldiio / librosa missing), so these rules are checked against their text, restated independently below."""
import os
import pickle

import numpy as np
import pytest
import torch


@pytest.fixture()
def corpus(tmp_path):
    rng = np.random.default_rng(0)
    lens = {"spk1_a": 57, "spk1_b": 20, "spk2_a": 19, "spk2_b": 133}
    with open(tmp_path / "feats.scp", "w") as fs, open(tmp_path / "len.scp", "w") as ls:
        for k, n in lens.items():
            feat = rng.normal(size=(n, 8)).astype(np.float32) * 3 + 1
            np.save(tmp_path / f"{k}.npy", feat)
            fs.write(f"{k} {tmp_path / (k + '.npy')}\n")   # "<seq> <path>", prepare_numpy_data.py:118
            ls.write(f"{k} {n}\n")                          # "<seq> <nframes>", :119
    return tmp_path, lens


def test_scp_segments_mvn_getitem(corpus):
    import datasets as D

    root, lens = corpus
    d = D.scp2dict(root / "len.scp", int)
    assert list(d.items()) == list(lens.items())
    ds = D.NumpyDataset(root / "feats.scp", root / "len.scp", min_len=20, mvn_path=str(root / "mvn.json"), seg_len=20, seg_shift=8)
    assert ds.seqlist == ["spk1_a", "spk1_b", "spk2_b"]          # min_len filter (datasets.py:82)
    assert len(ds) == 3                                            # number of sequences (datasets.py:138-139)
    want_nsegs = [(n - 20) // 8 + 1 for n in (57, 20, 133)]        # datasets.py:174
    assert ds.seq_nsegs == want_nsegs and ds.num_segments == sum(want_nsegs)
    assert [(s.seq, s.start, s.end) for s in ds.segs[:6]] == [("spk1_a", 0, 20), ("spk1_a", 8, 28), ("spk1_a", 16, 36),
                                                               ("spk1_a", 24, 44), ("spk1_a", 32, 52), ("spk1_b", 0, 20)]
    allf = np.concatenate([np.load(root / f"{k}.npy") for k in ds.seqlist]).astype(np.float64)
    np.testing.assert_allclose(ds.mvn_params["mean"].reshape(-1), allf.mean(0), rtol=1e-5)
    np.testing.assert_allclose(ds.mvn_params["std"].reshape(-1), allf.std(0), rtol=1e-4)
    idx, feat, nsegs = ds[7]                                       # 6th.. segment: spk2_b starts at index 6
    assert idx == 2 and nsegs == want_nsegs[2] and feat.shape == (20, 8)
    raw = np.load(root / "spk2_b.npy")[8:28]
    np.testing.assert_allclose(feat, (raw - ds.mvn_params["mean"]) / ds.mvn_params["std"], rtol=1e-6)
    np.testing.assert_allclose(ds.undo_mvn(feat), raw, rtol=1e-4, atol=1e-5)
    ds2 = D.NumpyDataset(root / "feats.scp", root / "len.scp", min_len=20, mvn_path=str(root / "mvn.json"))  # reload from json
    np.testing.assert_allclose(ds2.mvn_params["mean"], ds.mvn_params["mean"])
    rs = D.make_segs(["a"], [100], 20, 8, rand_seg=True, rng=np.random.default_rng(1))[0]
    assert len(rs) == 11 and all(0 <= s.start <= 80 and s.end - s.start == 20 for s in rs)


def test_checkpoint_layout_roundtrip(tmp_path):
    import utils as U
    from fhvae import FHVAE
    from simple_fhvae import SimpleFHVAE

    torch.manual_seed(0)
    m = FHVAE(4 * 6, [8, 8], [8, 8], 4, 4, [8, 8], seg_len=4, num_seqs=5)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
    U.save_checkpoint(m, opt, [1, 2, 3, 4, 5], {"train_loss_results": {0: 1.0}}, "timit_np_fbank", 3, 3, -1.0, -2.0, str(tmp_path))
    f = tmp_path / "fhvae_timit_np_fbank_e3.tar"
    assert f.exists() and (tmp_path / "best_model_fhvae_timit_np_fbank_e3.tar").exists()   # utils.py:148-152
    ck = torch.load(f, weights_only=False)
    assert set(ck) == {"best_val_lb", "best_epoch", "epoch", "model_type", "model_params", "optimizer", "state_dict",
                       "summary_vals", "values", "model_kwargs"}                               # utils.py:131-145 (+1)
    assert ck["model_type"] == "fhvae" and ck["model_params"][1:] == ([8, 8], [8, 8], 4, 4, [8, 8])
    m2, values, optim_state, start_epoch, best_val_lb, summary = U.load_checkpoint_file(f, finetune=False)
    assert start_epoch == 4 and best_val_lb == -2.0 and summary == [1, 2, 3, 4, 5] and optim_state is not None
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # a reference-style checkpoint: 5 model_params, no mu2_table
    s = SimpleFHVAE(32, [16, 16], [16, 16], 16, 16, [16, 16])
    ref_ck = dict(ck, model_type="simple_fhvae", model_params=([16, 16], [16, 16], 16, 16, [16, 16]), state_dict=s.state_dict())
    del ref_ck["model_kwargs"]
    torch.save(ref_ck, tmp_path / "ref.tar")
    s2 = U.load_checkpoint_file(tmp_path / "ref.tar", finetune=True, input_size=32)[0]
    assert all(torch.equal(a, b) for a, b in zip(s.state_dict().values(), s2.state_dict().values()))
    with pytest.raises(ValueError):
        U.load_checkpoint_file(tmp_path / "ref.tar", finetune=True)

    U.save_args(tmp_path, {"a": 1})
    assert U.load_args(tmp_path) == {"a": 1}


def test_dataset_matches_independent_oracle(corpus):
    """The package's host-side dataset against oracle/data_ref.py (an independent restatement of datasets.py's rules):
    kept sequences, segment table, MVN statistics and every item."""
    import datasets as D
    from oracle.data_ref import CorpusRef

    root, _ = corpus
    ds = D.NumpyDataset(root / "feats.scp", root / "len.scp", min_len=20, mvn_path=str(root / "mvn_o.json"), seg_len=20, seg_shift=8)
    ref = CorpusRef(root / "feats.scp", root / "len.scp", min_len=20, seg_len=20, seg_shift=8, mvn=True)
    assert ds.seqlist == ref.keys and len(ds) == len(ref) and ds.num_segments == ref.num_segments
    assert ds.seq_nsegs == ref.nseg.tolist()
    assert [ds.seq2idx[s.seq] for s in ds.segs] == ref.seq_of.tolist() and [s.start for s in ds.segs] == ref.start.tolist()
    np.testing.assert_array_equal(np.asarray(ds.mvn_params["mean"]), ref.mean)
    np.testing.assert_array_equal(np.asarray(ds.mvn_params["std"]), ref.std)
    for i in range(ds.num_segments):
        a, b = ds[i], ref.item(i)
        assert a[0] == b[0] and a[2] == b[2]
        np.testing.assert_array_equal(a[1], b[1])


def test_reference_layout_checkpoint_fixture_loads(golden_dir):
    """tests/golden/ref_checkpoint_simple_tiny.tar is a checkpoint in the reference's exact layout (utils.py:131-146: 5-value
    model_params, plain state_dict, torch.optim.Adam state) written from the imported reference model by make_golden.py."""
    import utils as U

    f = os.path.join(golden_dir, "ref_checkpoint_simple_tiny.tar")
    ck = torch.load(f, weights_only=False)
    assert set(ck) == {"best_val_lb", "best_epoch", "epoch", "model_type", "model_params", "optimizer", "state_dict",
                       "summary_vals", "values"} and len(ck["model_params"]) == 5
    with pytest.raises(ValueError):
        U.load_checkpoint_file(f, finetune=False)  # the reference stores no input size (utils.py:135-141)
    m, values, optim_state, start_epoch, best_val_lb, summary = U.load_checkpoint_file(f, finetune=False, input_size=32)
    assert type(m).__name__ == "SimpleFHVAE" and m.mu2_table is None and start_epoch == 4 and summary is None
    assert values == ck["values"] and best_val_lb == ck["best_val_lb"]
    sd = m.state_dict()
    assert list(sd) == list(ck["state_dict"])  # same keys, same order
    for k, v in ck["state_dict"].items():
        assert torch.equal(sd[k], v), k
    # the reference's Adam state: 24 parameters in the group, moments for the 16 that had a gradient (decoder detached)
    assert len(optim_state["param_groups"][0]["params"]) == 24 and len(optim_state["state"]) == 16
    m2, _, o2, e2, _, _ = U.load_checkpoint_file(f, finetune=True, input_size=32)
    assert o2 is None and e2 is None
