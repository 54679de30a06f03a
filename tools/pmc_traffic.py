#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes of `bench.py` into per-launch HBM bytes for the step-cell kernels.

  (GPU box)  cd /tmp && export TMPDIR=/tmp
             rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d OUT/fetch -- python3 bench.py ... --no-graph --no-roofline --no-cpu-baseline
             rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d OUT/write -- python3 bench.py ... (same)
  (anywhere) python tools/pmc_traffic.py OUT/fetch OUT/write <dtype> <B>  -> merges into profiles/pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB-like units of the TCC_EA request
counters; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane), so the
read side is doubled; WRITE_SIZE reads exactly for 16-B streaming stores.  The cell kernels mix 16-B tile loads with
4-B epilogue accesses, for which the counters are uncalibrated: the result is an estimate, stated as such.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            n = r["Kernel_Name"]
            c = out.setdefault(n, [0, 0.0])
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    return out


def main():
    fetch_dir, write_dir, dtype, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    tname = {"bf16": "unsigned short", "f32": "float"}[dtype]
    for short in ("lstm_fwd_step_kernel", "lstm_bwd_step_kernel", "lstm_fwd_cluster_kernel", "lstm_bwd_cluster_kernel",
                  "lstm_fwd_ksplit_kernel", "lstm_bwd_ksplit_kernel", "lstm_bwd_layer_kernel", "lstm_bwd_layer_ks_kernel", "lstm_bwd_layer_rs_kernel", "lstm_fwd_wr_kernel", "proj_kernel", "wgrad_kernel",
                  "disc_mfma_kernel"):
        typed = "step_kernel" in short  # the persistent kernels are bf16 only (no element type in their names)
        f = [(k, v) for k, v in fe.items() if short in k and (tname in k or not typed)]
        w = [(k, v) for k, v in wr.items() if short in k and (tname in k or not typed)]
        if not f or not w:
            continue
        # every instantiation of the kernel (e.g. the per-layer backward with and without the fused from-above term) counts
        fn, fv = sum(v[0] for _, v in f), sum(v[1] for _, v in f)
        wn, wv = sum(v[0] for _, v in w), sum(v[1] for _, v in w)
        rd = 2.0 * fv / fn * 1024.0   # FETCH_SIZE in KiB, x2 gfx950 correction for wide coalesced reads
        wb = wv / wn * 1024.0
        res["%s_%s_B%d" % (short, dtype, B)] = {"hbm_bytes_per_launch": rd + wb, "read_bytes": rd, "write_bytes": wb,
                                                "launches_sampled": fn, "note": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as is"}
        print(short, dtype, B, "read %.2f MB write %.2f MB per launch" % (rd / 1e6, wb / 1e6))
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
