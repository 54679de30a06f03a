import csv, glob, sys, os, re
for d in sys.argv[1:]:
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (re.sub(r"\(.*", "", r["Kernel_Name"])[:70], r["Counter_Name"])
            c = out.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += float(r["Counter_Value"])
    for (n, cn), (c, v) in sorted(out.items(), key=lambda x: -x[1][1])[:8]:
        print("%-12s %6d launches  avg %10.1f KiB-units  %s" % (cn, c, v / c, n))
