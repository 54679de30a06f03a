#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes: python tools/pmc_table.py <kernel substring,...> <dir> [<dir> ...]
Every directory is one pass (one --pmc counter set, --kernel-trace --output-format csv).  Prints, per matching kernel, the
average per dispatch of every counter found, plus the derived ratios the guide names (MI355X_MICROARCH.md, rocprofv3 PMC slots):
WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY as shares of WAVE_CYCLES, LDS bank-conflict share, L2 hit rate."""
import csv, glob, os, re, sys

pats = sys.argv[1].split(",")
acc = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            n = re.sub(r"\(.*", "", n)
            n = re.sub(r"^void ", "", n)
            if not any(p in n for p in pats):
                continue
            c = acc.setdefault(n, {}).setdefault(r["Counter_Name"], [0, 0.0])
            c[0] += 1
            c[1] += float(r["Counter_Value"])
for n in sorted(acc):
    v = {k: a[1] / a[0] for k, a in acc[n].items()}
    print("%s  (%d dispatches sampled)" % (n[:100], max(a[0] for a in acc[n].values())))
    for k in sorted(v):
        print("    %-28s %16.1f" % (k, v[k]))
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if k in v:
                print("    %-28s %15.1f %% of SQ_WAVE_CYCLES" % (k + " share", 100.0 * v[k] / wc))
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
        print("    %-28s %15.1f %% of SQ_LDS_IDX_ACTIVE" % ("LDS bank-conflict share", 100.0 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        print("    %-28s %15.1f %%" % ("L2 hit rate", 100.0 * v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
    if "FETCH_SIZE" in v:
        print("    %-28s %13.1f MB per dispatch (x2 gfx950 correction applied)" % ("HBM read", 2.0 * v["FETCH_SIZE"] * 1024 / 1e6))
    if "WRITE_SIZE" in v:
        print("    %-28s %13.1f MB per dispatch" % ("HBM write", v["WRITE_SIZE"] * 1024 / 1e6))
