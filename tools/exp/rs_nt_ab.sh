#!/bin/bash
# HBM traffic of the partial-dh backward kernel (FETCH_SIZE / WRITE_SIZE passes of the default bench): run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--no-graph --no-roofline --no-cpu-baseline --no-alt --no-hbm --steps 6 --warmup 2"
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $R/bench.py $ARGS > /dev/null 2>&1
python3 $R/tools/pmc_table.py lstm_bwd_layer_rs_kernel /tmp/pf /tmp/pw | grep -E "HBM|dispatches"
