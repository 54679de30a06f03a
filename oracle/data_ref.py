"""CPU oracle for the on-disk format + segment sampler -- TEST INFRASTRUCTURE ONLY.

An independent numpy restatement of the rules of the reference's datasets.py, written from its text
(the reference module itself cannot be imported here: datasets.py:4,9 need torchaudio / kaldiio, which
are not installed -- an ordinary ModuleNotFoundError).  PARITY UNPINNED BY THE REFERENCE for this row:
there is no reference fixture for it; this file is the checker for both the package's host-side
`datasets.NumpyDataset` and the HIP sampler `fhvae_segment_gather` (tests/ only import it).

Rules restated (reference file:line):
  scp files      "<key> <value>" per line, split at the first run of blanks, file order kept (datasets.py:13-32)
  kept sequences those with len >= min_len, in feats.scp order (datasets.py:80-83)
  segments       per kept sequence of n frames: nseg = (n - seg_len) // seg_shift + 1 segments starting at
                 k * seg_shift, k = 0..nseg-1 (datasets.py:155-185, rand_seg=False)
  item i         (sequence index, frames [start, start+seg_len) of that sequence's .npy, nseg of the sequence)
                 (datasets.py:214-223), normalised as (feat - mean) / std when MVN is on (datasets.py:100-105)
  MVN            mean = sum_frames x / N, std = sqrt(sum_frames x^2 / N - mean^2) per feature over all kept
                 sequences (datasets.py:225-235)
"""
import numpy as np


def read_scp(path):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.rstrip()
            cut = 0
            while cut < len(line) and not line[cut].isspace():
                cut += 1
            rows.append((line[:cut], line[cut:].lstrip()))
    return rows


class CorpusRef:
    def __init__(self, feat_scp, len_scp, min_len=1, seg_len=20, seg_shift=8, mvn=False):
        paths = dict(read_scp(feat_scp))
        lens = {k: int(v) for k, v in read_scp(len_scp)}
        self.keys = [k for k, _ in read_scp(feat_scp) if lens[k] >= min_len]
        self.paths = [paths[k] for k in self.keys]
        self.lens = np.array([lens[k] for k in self.keys], dtype=np.int64)
        self.seg_len, self.seg_shift = seg_len, seg_shift
        self.nseg = (self.lens - seg_len) // seg_shift + 1
        self.seq_of = np.repeat(np.arange(len(self.keys)), self.nseg)            # segment -> sequence index
        first = np.concatenate([[0], np.cumsum(self.nseg)[:-1]])
        self.start = (np.arange(self.nseg.sum()) - np.repeat(first, self.nseg)) * seg_shift
        self.mean = self.std = None
        if mvn:
            n, s1, s2 = 0, 0.0, 0.0
            for p in self.paths:
                a = np.load(p)
                s1 = s1 + np.sum(a, axis=0, keepdims=True)
                s2 = s2 + np.sum(a ** 2, axis=0, keepdims=True)
                n += a.shape[0]
            self.mean = s1 / float(n)
            self.std = np.sqrt(s2 / float(n) - self.mean ** 2)

    def __len__(self):
        return len(self.keys)

    @property
    def num_segments(self):
        return int(self.nseg.sum())

    def item(self, i):
        s = int(self.seq_of[i])
        a = np.load(self.paths[s])[int(self.start[i]):int(self.start[i]) + self.seg_len]
        if self.mean is not None:
            a = (a - self.mean) / self.std
        return s, a, int(self.nseg[s])

    def batch(self, ids):
        items = [self.item(int(i)) for i in ids]
        return (np.array([t[0] for t in items], dtype=np.int64), np.stack([t[1] for t in items]).astype(np.float32),
                np.array([t[2] for t in items], dtype=np.int64))
