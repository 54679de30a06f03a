set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/rf
python3 $R/bench.py > $R/gpurun_out/rf/bench_default.json 2> $R/gpurun_out/rf/bench_default.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt2048 -o r -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-graph --steps 10 --warmup 3 > $R/gpurun_out/rf/prof2048.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py ... --no-graph --steps 10 --warmup 3 (per-kernel totals via tools/rocpd_stats.py; 13 steps traced, times per step)"; python3 $R/tools/rocpd_stats.py $(find /tmp/kt2048 -name "*.db" | head -1) 13; } > $R/gpurun_out/rf/B2048_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/kt256 -o r -- python3 $R/bench.py --batch 256 --no-cpu-baseline --no-roofline --no-graph --steps 10 --warmup 3 > $R/gpurun_out/rf/prof256.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py ... --no-graph --steps 10 --warmup 3 (per-kernel totals via tools/rocpd_stats.py; 13 steps traced, times per step)"; python3 $R/tools/rocpd_stats.py $(find /tmp/kt256 -name "*.db" | head -1) 13; } > $R/gpurun_out/rf/B256_kernel_stats.txt
echo refreshed
