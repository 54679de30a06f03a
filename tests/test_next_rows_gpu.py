"""GPU tests of the SURVEY 8f rows built on HIP kernels: resident-pool segment sampler (fhvae_segment_gather),
closed-form mu2 estimate (fhvae_mu2_accumulate/_finalize, utils.estimate_mu2_dict), latent extraction (encode)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from test_data_ckpt_cpu import corpus  # noqa: F401
from test_ops_gpu import close, dev, hb  # noqa: F401


def test_resident_pool_matches_dataset_getitem(hb, corpus):
    import datasets as D

    root, _ = corpus
    ds = D.NumpyDataset(root / "feats.scp", root / "len.scp", min_len=20, mvn_path=str(root / "mvn.json"), seg_len=20, seg_shift=8)
    pool = D.ResidentSegmentPool(ds)
    assert len(pool) == ds.num_segments and pool.num_seqs == 3
    ids = torch.tensor([0, 7, 5, len(pool) - 1, 7], device="cuda")
    idxs, x, nsegs = pool.batch(ids)
    for k, i in enumerate(ids.tolist()):
        si, feat, ns = ds[i]
        assert idxs[k].item() == si and nsegs[k].item() == ns
        close(x[k], torch.from_numpy(np.asarray(feat, dtype=np.float32)), rtol=1e-5, what="segment %d" % i)
    x2, x_tm = hb.segment_gather(pool.pool, pool.seg_start[ids], 20, pool.mean, pool.inv_std, time_major=True)
    close(x_tm.transpose(0, 1), x2, rtol=0, what="time-major")
    seen = sum(b[1].shape[0] for b in pool.epoch(4, shuffle=True))
    assert seen == len(pool)


def test_segment_gather_matches_independent_oracle(hb, corpus):
    """fhvae_segment_gather (through ResidentSegmentPool) against oracle/data_ref.py -- NOT against the package's own
    NumpyDataset: indices, features (MVN fused in the kernel: (x - mean) * (1/std), 1 ulp of the divide form) and counts."""
    import datasets as D
    from oracle.data_ref import CorpusRef

    root, _ = corpus
    for mvn in (True, False):
        ds = D.NumpyDataset(root / "feats.scp", root / "len.scp", min_len=20, mvn_path=str(root / "mvn_g.json") if mvn else None,
                            seg_len=20, seg_shift=8)
        ref = CorpusRef(root / "feats.scp", root / "len.scp", min_len=20, seg_len=20, seg_shift=8, mvn=mvn)
        pool = D.ResidentSegmentPool(ds)
        assert len(pool) == ref.num_segments
        ids = torch.arange(len(pool) - 1, -1, -1, device="cuda")  # every segment, reversed order
        idxs, x, nsegs = pool.batch(ids)
        want_i, want_x, want_n = ref.batch(ids.tolist())
        assert idxs.cpu().tolist() == want_i.tolist() and nsegs.cpu().tolist() == want_n.tolist()
        if mvn:
            close(x, torch.from_numpy(want_x), rtol=2e-6, what="segments (mvn)")
        else:
            assert torch.equal(x.cpu(), torch.from_numpy(want_x)), "raw segments must be bit-exact copies"
        # an epoch covers every segment exactly once
        seen = torch.cat([b[0] for b in pool.epoch(5, shuffle=True)]).cpu()
        assert sorted(seen.tolist()) == sorted(want_i.tolist())


def test_mu2_estimate_and_estimate_mu2_dict(hb):
    import utils as U
    from fhvae import FHVAE

    torch.manual_seed(0)
    N, S, D = 500, 37, 16
    z = torch.randn(N, D)
    idx = torch.randint(0, S - 3, (N,))  # the last sequences never occur
    want, cnt = R.estimate_mu2(z, idx, S)
    est = hb.Mu2Estimator(S, D, "cuda")
    est.add(dev(z[:200]), dev(idx[:200]))
    est.add(dev(z[200:]), dev(idx[200:]))
    got, gc = est.result(0.25)
    close(got, want, what="mu2")
    close(gc, cnt, rtol=0, what="counts")
    # drop-in: estimate_mu2_dict(model, loader, num_seqs) -> {y: mu2_y}  (utils.py:45-60)
    T, F, H = 20, 80, 32
    m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], num_seqs=S).cuda()
    ref = R.FHVAERef(T * F, [H, H], [H, H], D, D, [H, H])
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items() if k != "mu2_table"})
    xs = [torch.randn(24, T, F) for _ in range(3)]
    ys = [torch.randint(0, 6, (24,)) for _ in range(3)]
    d = U.estimate_mu2_dict(m, [(y, x.cuda(), torch.full((24,), 9)) for x, y in zip(xs, ys)], S)
    with torch.no_grad():  # oracle: z2_mu of every segment, then the closed form
        z2 = []
        for x in xs:
            _, (h_n, _) = ref.z2_pre_encoder.lstm(x)
            z2.append(ref.z2_gauss_layer(ref._final_h(h_n))[0])
    want, cnt = R.estimate_mu2(torch.cat(z2), torch.cat(ys), S)
    assert sorted(d) == torch.nonzero(cnt > 0).flatten().tolist()
    for y, v in d.items():
        close(v, want[y], rtol=2e-4, what="mu2_dict[%d]" % y)


def test_encode_matches_oracle(hb):
    from simple_fhvae import SimpleFHVAE

    torch.manual_seed(2)
    m = SimpleFHVAE(4 * 8, [16, 16], [16, 16], 16, 16, [16, 16]).cuda()
    ref = R.SimpleFHVAERef(4 * 8, [16, 16], [16, 16], 16, 16, [16, 16])
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    x = torch.randn(9, 4, 8)
    with torch.no_grad():
        z1, z2 = m.encode(x.cuda())
        xf = x.reshape(9, -1)
        z2_ref = ref.z2_gauss_layer(ref.z2_pre_encoder(xf))[0]
        z1_ref = ref.z1_gauss_layer(ref.z1_pre_encoder(torch.cat([xf, z2_ref], -1)))[0]
    close(z2, z2_ref, what="z2_mu")
    close(z1, z1_ref, what="z1_mu")


def test_train_model_main_real_scp_and_synthetic(hb, corpus, tmp_path, capsys):
    """The train_model-shaped loop end to end on the GPU: scp/npy corpus through the resident pool, hierarchical mu2
    initialisation, reference-layout checkpoints; then the synthetic default with SimpleFHVAE."""
    import train_model as TM
    import utils as U

    root, _ = corpus
    exp = tmp_path / "exp"
    rc = TM.main(["--train-feat-scp", str(root / "feats.scp"), "--train-len-scp", str(root / "len.scp"), "--mvn-path",
                  str(root / "mvn2.json"), "--z1-hus", "16", "16", "--z2-hus", "16", "16", "--x-hus", "16", "16", "--z1-dim", "8",
                  "--z2-dim", "8", "--epochs", "2", "--training-batch-size", "8", "--exp-dir", str(exp), "--hierarchical",
                  ])
    out = capsys.readouterr().out
    assert rc == 0 and "Training complete!" in out and "hierarchical: mu2 re-estimated for 3 of 3" in out
    m = U.load_checkpoint_file(exp / "fhvae_run_e1.tar", finetune=True)[0]
    assert m.mu2_table.shape == (3, 8) and m.n_feat == 8
    rc = TM.main(["--model-type", "simple_fhvae", "--epochs", "1", "--train-segments", "64", "--dev-segments", "32",
                  "--training-batch-size", "32", "--num-seqs", "10"])
    assert rc == 0
