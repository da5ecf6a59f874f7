#!/bin/bash
# Evidence of a round, collected on the GPU box (run through gpurun from the repository root), in three parts that each fit one call:
#   tools/profile_round.sh r05 a    the default bench line, plain and under the kernel trace (per-kernel averages by workload)
#   tools/profile_round.sh r05 b    HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the LLR workloads:
#                                   headline, Pareto lengths, the preset's default window, int16 rows
#   tools/profile_round.sh r05 c    the CNN step at the 200 k window: kernel stats, matrix-core counters, traffic; the flip census
# Everything lands under gpurun_out/<tag>_summary/; the summaries worth keeping are copied to profiles/ by hand afterwards
# (profiles/summarize_pmc.py, profiles/summarize_sq.py).  Counter passes run on their own (no --stats / trace domains with --pmc).
set -e
TAG=${1:-r05}
PART=${2:-a}
export TMPDIR=/tmp
OUT=gpurun_out
S=$OUT/${TAG}_summary
mkdir -p $S
pmc() { # pmc <name> <reads per launch> <T> <bench arguments ...>: FETCH_SIZE and WRITE_SIZE passes + their summary
  local name=$1 reads=$2 T=$3; shift 3
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_${name}_f -o run --output-format csv -- python3 bench.py "$@" > /dev/null 2> $OUT/${TAG}_${name}_f.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_${name}_w -o run --output-format csv -- python3 bench.py "$@" > /dev/null 2> $OUT/${TAG}_${name}_w.err
  python3 profiles/summarize_pmc.py $OUT/${TAG}_${name}_f/run_counter_collection.csv $OUT/${TAG}_${name}_w/run_counter_collection.csv $reads $T $S/${name}.json > $S/${name}.txt
  rm -rf $OUT/${TAG}_${name}_f $OUT/${TAG}_${name}_w
}
case $PART in
a)
  BENCH="python3 bench.py --steps 6 --warmup 2"
  $BENCH > $S/bench.json 2> $OUT/${TAG}_bench.err
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run -- $BENCH --cpu-sample 0 > $S/bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err
  python3 profiles/summarize_phases.py $OUT/${TAG}_stats/run_results.db $S/kernel_stats_by_workload.csv > $S/kernel_stats_by_workload.txt
  rm -rf $OUT/${TAG}_stats
  ;;
b)
  pmc traffic 96000 200000 --steps 2 --warmup 1 --no-secondary --cpu-sample 0
  pmc traffic_pareto 96000 200000 --lens pareto --steps 2 --warmup 1 --no-secondary --cpu-sample 0
  pmc traffic_default_window 96000 16000 --max_obs_trace 16000 --steps 2 --warmup 1 --no-secondary --cpu-sample 0
  pmc traffic_int16 96000 200000 --int16 --steps 2 --warmup 1 --no-secondary --cpu-sample 0
  ;;
c)
  CNN="python3 bench.py --primary cnn --reads 8000 --steps 3 --warmup 1 --no-secondary --cpu-sample 0"
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_cnn_stats -o run -- $CNN > $S/cnn_under_rocprof.json 2> $OUT/${TAG}_cnn_under_rocprof.err
  python3 profiles/summarize_pmc.py --stats $OUT/${TAG}_cnn_stats/run_results.db $S/cnn200k_kernel_stats.csv > $S/cnn200k_kernel_stats.txt
  rm -rf $OUT/${TAG}_cnn_stats
  SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
  rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE -d $OUT/${TAG}_cnn_sq -o run --output-format csv -- $CNN > /dev/null 2> $OUT/${TAG}_cnn_sq.err || { rm -rf $OUT/${TAG}_cnn_sq; rocprofv3 --kernel-trace --pmc $SQ -d $OUT/${TAG}_cnn_sq -o run --output-format csv -- $CNN > /dev/null 2>> $OUT/${TAG}_cnn_sq.err; }
  python3 profiles/summarize_sq.py $OUT/${TAG}_cnn_sq/run_counter_collection.csv $S/sq_counters_cnn200k.json > $S/sq_counters_cnn200k.txt
  rm -rf $OUT/${TAG}_cnn_sq
  pmc cnn200k_traffic 8000 200000 --primary cnn --reads 8000 --steps 3 --warmup 1 --no-secondary --cpu-sample 0
  # the conv stacks' flip census against the reference's own cnn_detect (tests/test_gpu_cnn.py keeps its report in the test's tmp dir)
  ADP_FLIPS_REPORT_DIR=$PWD/$S python3 -m pytest tests/test_gpu_cnn.py -q -m gpu -k flips > $S/flips_pytest.txt 2>&1 || true
  ;;
esac
ls $S
