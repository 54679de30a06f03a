"""Scan the gfx950 ISA of every kernel of csrc/*.hip for MFMAs that wait for the fragment they need right behind its read:
   ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma
A software pipeline written in the source can come out of hipcc like that (one LDS latency per product) when the kernel sits near
the register limit of its occupancy: lstm_fwd_wr.hip's products did until the order was pinned with
__builtin_amdgcn_sched_barrier(0).  No GPU needed: python tools/isa_serial_scan.py [file.hip ...]"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch-scalablefhvae_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I" + os.path.join(ROOT, "include"),
         "--cuda-device-only", "-S"]


def scan(asm_path):
    kern, stats, prev = None, {}, []
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, prev = m.group(1), []
            stats[kern] = [0, 0]
            continue
        t = line.strip()
        if kern is None or not t or t[0] in ";.":
            continue
        if t.startswith("v_mfma"):
            stats[kern][1] += 1
            if len(prev) >= 2 and prev[-1].startswith("s_waitcnt lgkmcnt(0)") and prev[-2].startswith("ds_read"):
                stats[kern][0] += 1
        prev = (prev + [t])[-3:]
    return stats


def main():
    files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            out = os.path.join(tmp, os.path.basename(f) + ".s")
            r = subprocess.run([hipcc] + FLAGS + [f, "-o", out], capture_output=True, text=True, cwd=CSRC)
            if r.returncode != 0 or not os.path.exists(out):
                print("%s: did not compile\n%s" % (f, r.stderr[-400:]))
                continue
            rows = [(a, b, k) for k, (a, b) in scan(out).items() if b]
            if rows:
                print("== %s" % os.path.basename(f))
                for a, b, k in sorted(rows, reverse=True)[:12]:
                    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
                    print("  %4d of %4d MFMAs right behind their fragment's read   %s" % (a, b, name[:110]))


if __name__ == "__main__":
    main()
