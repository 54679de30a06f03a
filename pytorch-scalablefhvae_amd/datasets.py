"""datasets.py -- the reference's on-disk format and segmenting rules (datasets.py:13-235) plus a MI355X-native
sampler that keeps a whole split resident in HBM.

Kept from the reference (same names, constructor arguments and return values):
  scp2dict (:13-32); Segment (:35-47); NumpyDataset(feat_scp, len_scp, min_len, mvn_path, seg_len, seg_shift, rand_seg)
  with `__getitem__(i) -> (seq_idx, feat (seg_len, F) float ndarray, nsegs)` (:214-223), `_make_segs`
  (`nseg = (len - seg_len) // seg_shift + 1`, :155-185), MVN statistics over the kept sequences (:225-235) and
  `apply_mvn` / `undo_mvn` (:100-105, :131-136).  `len(dataset)` is the number of SEQUENCES, as in the reference
  (:138-139) -- the train loop passes it to the model as `num_seqs` (train_model.py:448); `num_segments` is new.
Not reproduced: `json.dump` of ndarrays in `_mvn_prep` (:111-112 raises TypeError; lists are written instead);
KaldiDataset (needs the external kaldiio package: out of scope).

New: `ResidentSegmentPool` loads every kept utterance once, concatenates them into one (frames, F) f32 tensor in HBM
(288 GB hold ~900 M frames of 80-bin features) and cuts minibatches with the `fhvae_segment_gather` kernel
(MVN fused), so the training step has no DataLoader, no worker processes and no per-step H2D copy.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch


def scp2dict(path, dtype=str, seqlist=None):
    """scp file ("<key> <value>" per line, split at the first blank run) -> OrderedDict in file order, values cast with
    `dtype`, optionally restricted to the keys in `seqlist` (contract of datasets.py:13-32)."""
    wanted = None if seqlist is None else set(seqlist)
    out = OrderedDict()
    with open(path) as fh:
        for raw in fh:
            key, value = raw.rstrip().split(None, 1)
            if wanted is None or key in wanted:
                out[key] = dtype(value)
    return out


class Segment:
    """Frames [start, end) of utterance `seq` (the record type of datasets.py:35-47; printed as "seq, start, end")."""

    __slots__ = ("seq", "start", "end")

    def __init__(self, seq, start, end):
        self.seq, self.start, self.end = seq, start, end

    def __repr__(self):
        return "%s, %s, %s" % (self.seq, self.start, self.end)

    __str__ = __repr__


def make_segs(seqs, lens, seg_len=20, seg_shift=8, rand_seg=False, rng=None):
    """datasets.py:155-185.  Returns (segments, per-sequence segment counts)."""
    segs, nsegs = [], []
    rng = rng if rng is not None else np.random
    for seq, l in zip(seqs, lens):
        nseg = (l - seg_len) // seg_shift + 1
        nsegs.append(nseg)
        if rand_seg:
            starts = rng.choice(range(l - seg_len + 1), nseg)
        else:
            starts = np.arange(nseg) * seg_shift
        for start in starts:
            segs.append(Segment(seq, int(start), int(start) + seg_len))
    return segs, nsegs


class NumpyDataset(torch.utils.data.Dataset):
    def __init__(self, feat_scp: Path, len_scp: Path, min_len: int = 1, mvn_path: str = None, seg_len: int = 20,
                 seg_shift: int = 8, rand_seg: bool = False, sequence_list=None):
        every_feat = scp2dict(feat_scp)
        every_len = scp2dict(len_scp, int, every_feat.keys())
        self.seg_len, self.seg_shift, self.rand_seg = seg_len, seg_shift, rand_seg
        # utterances kept: an explicit list, or all that are at least min_len frames long (datasets.py:80-83)
        self.seqlist = list(sequence_list) if sequence_list is not None else [k for k, n in every_len.items() if n >= min_len]
        self.feats = OrderedDict((k, every_feat[k]) for k in self.seqlist)
        self.lens = OrderedDict((k, every_len[k]) for k in self.seqlist)
        print("%s: %d out of %d kept, min_len = %s" % (type(self).__name__, len(self.feats), len(every_feat), min_len))
        self.seq_keys = list(self.seqlist)
        self.seq_feats = list(self.feats.values())
        self.seq_lens = list(self.lens.values())
        self.segs, self.seq_nsegs = make_segs(self.seq_keys, self.seq_lens, seg_len, seg_shift, rand_seg)
        self.seq2idx = {seq: i for i, seq in enumerate(self.seq_keys)}
        self._mvn_prep(mvn_path)

    # -- mean / variance normalisation (datasets.py:100-136, :225-235) -----------------------------
    def _compute_mvn(self):
        """Per-feature mean and standard deviation over all frames of the kept utterances: first and second moments are
        summed utterance by utterance in the arrays' own dtype, then std = sqrt(E[x^2] - mean^2) (datasets.py:225-235)."""
        frames, s1, s2 = 0.0, 0.0, 0.0
        for path in self.feats.values():
            a = np.load(path)
            s1 = s1 + a.sum(axis=0, keepdims=True)
            s2 = s2 + np.square(a).sum(axis=0, keepdims=True)
            frames += a.shape[0]
        mean = s1 / frames
        return {"mean": mean, "std": np.sqrt(s2 / frames - np.square(mean))}

    def _mvn_prep(self, mvn_path):
        self.mvn_params = None
        if mvn_path is None:
            return
        if os.path.exists(mvn_path):
            with open(mvn_path) as f:
                self.mvn_params = {k: np.asarray(v) for k, v in json.load(f).items()}
        else:
            self.mvn_params = self._compute_mvn()
            with open(mvn_path, "w") as f:  # (the reference hands ndarrays to json.dump, datasets.py:111-112: TypeError)
                json.dump({k: np.asarray(v).tolist() for k, v in self.mvn_params.items()}, f)

    def apply_mvn(self, feats):
        p = self.mvn_params
        return feats if p is None else (feats - p["mean"]) / p["std"]

    def undo_mvn(self, feats):
        p = self.mvn_params
        return feats if p is None else feats * p["std"] + p["mean"]

    def __len__(self):
        return len(self.seqlist)  # number of sequences (datasets.py:138-139): the loop's num_seqs

    @property
    def num_segments(self):
        return len(self.segs)

    def __getitem__(self, index):
        """Returns sequence index, feature (seg_len, F) and the sequence's number of segments (datasets.py:214-223)."""
        seg = self.segs[index]
        idx = self.seq2idx[seg.seq]
        with open(self.seq_feats[idx], "rb") as f:
            feat = np.load(f)[seg.start:seg.end]
        feat = self.apply_mvn(feat)
        return idx, feat, self.seq_nsegs[idx]


class ResidentSegmentPool:
    """All utterances of a NumpyDataset resident in HBM + device-side minibatch cutting (fhvae_segment_gather)."""

    def __init__(self, dataset: NumpyDataset, device="cuda"):
        import hip_binding as hb

        self.hb = hb
        self.T = dataset.seg_len
        feats = [np.load(p).astype(np.float32) for p in dataset.seq_feats]
        offs = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
        self.pool = torch.from_numpy(np.concatenate(feats, axis=0)).to(device)  # (frames, F), one H2D copy per split
        self.num_seqs = len(dataset)
        seq_of = np.array([dataset.seq2idx[s.seq] for s in dataset.segs], dtype=np.int64)
        self.seg_start = torch.from_numpy(offs[seq_of] + np.array([s.start for s in dataset.segs], dtype=np.int64)).to(device)
        self.seg_seq = torch.from_numpy(seq_of).to(device)
        self.seg_nsegs = torch.from_numpy(np.array(dataset.seq_nsegs, dtype=np.int64)[seq_of]).to(device)
        if dataset.mvn_params is not None:
            self.mean = torch.from_numpy(np.asarray(dataset.mvn_params["mean"], dtype=np.float32).reshape(-1)).to(device)
            self.inv_std = torch.from_numpy((1.0 / np.asarray(dataset.mvn_params["std"], dtype=np.float64)).astype(np.float32).reshape(-1)).to(device)
        else:
            self.mean = self.inv_std = None

    def __len__(self):
        return self.seg_start.shape[0]

    def batch(self, seg_ids: torch.Tensor):
        """seg_ids (B,) int64 on the device -> (idxs (B,), features (B,T,F), nsegs (B,)) like the reference's collate."""
        st = self.seg_start[seg_ids]
        x = self.hb.segment_gather(self.pool, st, self.T, self.mean, self.inv_std)
        return self.seg_seq[seg_ids], x, self.seg_nsegs[seg_ids]

    def epoch(self, batch_size: int, shuffle=True, generator=None, drop_last=False):
        n = len(self)
        order = torch.randperm(n, device=self.seg_start.device, generator=generator) if shuffle else torch.arange(n, device=self.seg_start.device)
        for s in range(0, n, batch_size):
            ids = order[s:s + batch_size]
            if drop_last and ids.shape[0] < batch_size:
                break
            yield self.batch(ids)
