#!/bin/bash
# c5 counter table (MfmaUtil, SQ wait/busy, instruction mix) of the K5 / cell / f32 weight-gradient kernels; run on the GPU box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--config c5 --no-graph --no-roofline --no-cpu-baseline --no-alt --no-hbm --steps 3 --warmup 1"
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d /tmp/p1 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p2 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/p3 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
python3 $R/tools/pmc_table.py disc_mfma_kernel,cell_fwd_kernel,cell_bwd_kernel,wgrad_f32_kernel /tmp/p1 /tmp/p2 /tmp/p3 > $R/gpurun_out/pmc_c5.txt 2>&1
grep -E "dispatches|share|MfmaUtil|INSTS" $R/gpurun_out/pmc_c5.txt
