#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE) and matrix-pipe / LDS counters of the hot kernels, separate --pmc passes (no trace domains
# combined with --pmc beyond --kernel-trace).  Run on the GPU box: bash tools/collect_pmc.sh ; outputs under gpurun_out/pmc/
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-graph --no-roofline --no-cpu-baseline --steps 6 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
python3 - <<PY
import sys, shutil, os
sys.argv = ["pmc_traffic.py", "/tmp/pmc_fetch", "/tmp/pmc_write", "bf16", "2048"]
sys.path.insert(0, "$R/tools")
import runpy
runpy.run_path("$R/tools/pmc_traffic.py", run_name="__main__")
shutil.copy("$R/profiles/pmc_traffic.json", "$O/pmc_traffic.json")
PY
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python3 $R/bench.py $ARGS > $O/mfma.log 2>&1
python3 $R/tools/pmc_quick.py /tmp/pmc_mfma > $O/mfma_util.txt 2>&1 || true
echo pmc done
