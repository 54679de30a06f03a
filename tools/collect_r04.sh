#!/bin/bash
# Round-4 evidence for profiles/: run on the GPU box (bash tools/collect_r04.sh [what...]); outputs under gpurun_out/r04/.
#   stats   rocprofv3 --kernel-trace per-kernel totals of the default bench (c3, B=2048, bf16) and of c2 B=256, c4, c5
#   pmc     separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MfmaUtil) of the default bench -> pmc_traffic.json
#   bench   the bench lines themselves (default, c4, c5, zipf)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
WHAT=${@:-stats bench}
cd /tmp && export TMPDIR=/tmp
PROF="--no-cpu-baseline --no-roofline --no-alt --no-graph --steps 10 --warmup 3"
stats() {  # name, bench args
  local n=$1; shift
  rm -rf /tmp/kt_$n
  rocprofv3 --kernel-trace -d /tmp/kt_$n -o r -- python3 $R/bench.py $PROF "$@" > $O/prof_$n.log 2>&1
  db=$(find /tmp/kt_$n -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace -- python3 bench.py $PROF $* (per-kernel totals via tools/rocpd_stats.py; 13 steps traced, times per step)"; python3 $R/tools/rocpd_stats.py $db 13 60; } > $O/${n}_kernel_stats.txt
  echo stats $n done
}
for w in $WHAT; do
  case $w in
    stats)
      stats c3_bf16_B2048
      stats c2_bf16_B256 --config c2 --batch 256
      stats c4 --config c4
      stats c5 --config c5
      ;;
    bench)
      python3 $R/bench.py > $O/bench_default_c3.json 2> $O/bench_default_c3.log
      python3 $R/bench.py --config c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.log
      python3 $R/bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.log
      python3 $R/bench.py --idx zipf --no-cpu-baseline --no-alt > $O/bench_c3_zipf.json 2> $O/bench_c3_zipf.log
      python3 $R/bench.py --config c1 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.log
      echo bench done
      ;;
    pmc)
      ARGS="--no-graph --no-roofline --no-cpu-baseline --no-alt --steps 6 --warmup 2"
      rm -rf /tmp/pmc_fetch /tmp/pmc_write /tmp/pmc_mfma
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py $ARGS > $O/pmc_fetch.log 2>&1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py $ARGS > $O/pmc_write.log 2>&1
      python3 $R/tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write bf16 2048 > $O/pmc_traffic.txt 2>&1
      cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
      rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python3 $R/bench.py $ARGS > $O/pmc_mfma.log 2>&1
      python3 $R/tools/pmc_quick.py /tmp/pmc_mfma > $O/pmc_mfma_util.txt 2>&1 || true
      python3 $R/tools/pmc_quick.py /tmp/pmc_fetch /tmp/pmc_write > $O/pmc_fetch_write_top.txt 2>&1 || true
      echo pmc done
      ;;
    stream)  # achieved HBM GB/s of the streaming kernels: HIP-event table + the rocprofv3 durations of the same launches
      rm -rf /tmp/kt_stream
      rocprofv3 --kernel-trace -d /tmp/kt_stream -o r -- python3 $R/tools/bench_stream.py > $O/stream_events.txt 2> $O/stream.log
      { grep -v "^W2026\|^E2026\|amdgpu.ids" $O/stream_events.txt; echo; echo "# rocprofv3 --kernel-trace -- python3 tools/bench_stream.py: average kernel durations of the same launches"; python3 $R/tools/rocpd_stats.py $(find /tmp/kt_stream -name "*.db" | head -1) 1 14; } > $O/stream_kernels_hbm.txt
      echo stream done
      ;;
    dist)  # the distributed runner on one rank (captured step) beside the single-GPU step
      python3 $R/bench.py --force-dist --no-cpu-baseline --no-alt --no-hbm > $O/bench_c3_dist_world1.json 2> $O/bench_c3_dist_world1.log
      python3 $R/bench.py --no-cpu-baseline --no-alt --no-hbm --no-roofline > $O/bench_c3_single_ref.json 2> $O/bench_c3_single_ref.log
      echo dist done
      ;;
    stall)  # stall attribution of the recurrent kernels and the GEMMs around them: separate SQ / TCC passes (8 SQ slots, 4 TCC slots per pass)
      ARGS="--no-graph --no-roofline --no-cpu-baseline --no-alt --steps 6 --warmup 2"
      K="lstm_bwd_layer_rs_kernel,lstm_fwd_wr_kernel,proj_kernel,wgrad_kernel,disc_lp_kernel,elbo_bwd_pair_kernel"
      rm -rf /tmp/pmc_s1 /tmp/pmc_s2 /tmp/pmc_s3 /tmp/pmc_s4
      rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_s1 -- python3 $R/bench.py $ARGS > $O/pmc_s1.log 2>&1
      rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_s2 -- python3 $R/bench.py $ARGS > $O/pmc_s2.log 2>&1
      rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pmc_s3 -- python3 $R/bench.py $ARGS > $O/pmc_s3.log 2>&1
      rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d /tmp/pmc_s4 -- python3 $R/bench.py $ARGS > $O/pmc_s4.log 2>&1
      python3 $R/tools/pmc_table.py $K /tmp/pmc_s1 /tmp/pmc_s2 /tmp/pmc_s3 /tmp/pmc_s4 /tmp/pmc_fetch /tmp/pmc_write /tmp/pmc_mfma > $O/pmc_stall_table.txt 2>&1 || true
      tail -3 $O/pmc_s1.log $O/pmc_s2.log $O/pmc_s3.log $O/pmc_s4.log > $O/pmc_stall_logs_tail.txt 2>&1 || true
      echo stall done
      ;;
  esac
done
