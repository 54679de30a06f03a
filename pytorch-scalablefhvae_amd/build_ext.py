"""Build libfhvae_hip.so (gfx950) from csrc/*.hip with hipcc -- in-tree, no torch extension machinery.

    python pytorch-scalablefhvae_amd/build_ext.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libfhvae_hip.so")
SOURCES = ["gemm.hip", "lstm.hip", "lstm_cluster.hip", "lstm_bwd_rs.hip", "lstm_fwd_wr.hip", "proj.hip", "loss.hip", "disc_mfma.hip", "data.hip", "trace.hip", "wgrad.hip", "wgrad_f32.hip", "disc_lp.hip", "lstm_cell.hip"]
# -amdgpu-mfma-vgpr-form: MFMA results stay in VGPRs.  By default the backend puts accumulators in AGPRs and copies them
# around every non-MFMA use (v_accvgpr_read/write, 4 issue cycles each): 288 such moves in the long-K GEMM's loop, none with
# this form; measured 791 -> 810 k segments/s at B=2048, 275 -> 284 k at B=256.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form"]
EXTRA_FLAGS = {}  # per-file additions


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "fhvae_hip.h"))
    return hdrs


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _deps()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + os.environ.get("FHVAE_EXTRA_HIPCC", "").split() + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        return obj

    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or force or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
    if verbose:
        print("libfhvae_hip.so: %d translation unit(s) rebuilt -> %s" % (len(jobs), LIB))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
