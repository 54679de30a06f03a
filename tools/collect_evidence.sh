#!/bin/bash
# Round evidence for profiles/: achieved HBM GB/s of the streaming kernels, the discriminative kernel's FLOP rate, and
# per-kernel rocprofv3 totals of configs c3..c5 on one GPU.  Run on the GPU box: bash tools/collect_evidence.sh
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/ev
python3 $R/tools/bench_stream.py > $R/gpurun_out/ev/stream.txt 2>&1
python3 $R/tools/bench_disc.py > $R/gpurun_out/ev/disc.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in c3 c4 c5; do
  python3 $R/bench.py --config $c --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/ev/bench_$c.json 2> $R/gpurun_out/ev/bench_$c.log
  rocprofv3 --kernel-trace -d /tmp/kt_$c -o r -- python3 $R/bench.py --config $c --no-cpu-baseline --no-roofline --no-graph --steps 10 --warmup 3 > $R/gpurun_out/ev/prof_$c.log 2>&1
  db=$(find /tmp/kt_$c -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py ... --no-graph --steps 10 --warmup 3 (per-kernel totals via tools/rocpd_stats.py; 13 steps traced, times per step)"; python3 $R/tools/rocpd_stats.py $db 13; } > $R/gpurun_out/ev/${c}_kernel_stats.txt
  echo done $c
done
