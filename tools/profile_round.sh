#!/bin/bash
# Evidence of a round, collected on the GPU box (run through gpurun from the repository root):
#   tools/profile_round.sh r05
# writes everything under gpurun_out/<tag>_*; the summaries worth keeping are copied to profiles/ by hand afterwards
# (profiles/summarize_pmc.py, profiles/summarize_sq.py).  Counter passes run on their own (no --stats / trace domains with --pmc).
set -e
TAG=${1:-r05}
export TMPDIR=/tmp
OUT=gpurun_out
BENCH="python3 bench.py --steps 6 --warmup 2"
# 1. the default run (headline + secondaries + CPU baselines), plain
$BENCH > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# 2. the same command under the kernel trace (per-kernel average durations)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run -- $BENCH --cpu-sample 0 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err
# 3. HBM traffic of the headline workload at its real launch size (96 000 reads): FETCH_SIZE and WRITE_SIZE in separate passes
LLR="python3 bench.py --steps 2 --warmup 1 --no-secondary --cpu-sample 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch -o run --output-format csv -- $LLR > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write -o run --output-format csv -- $LLR > /dev/null 2> $OUT/${TAG}_pmc_write.err
# 4. the CNN step at the 200 k window: kernel stats, matrix-core counters, traffic
CNN="python3 bench.py --primary cnn --reads 8000 --steps 3 --warmup 1 --no-secondary --cpu-sample 0"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_cnn_stats -o run -- $CNN > $OUT/${TAG}_cnn_under_rocprof.json 2> $OUT/${TAG}_cnn_under_rocprof.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/${TAG}_cnn_pmc_sq -o run --output-format csv -- $CNN > /dev/null 2> $OUT/${TAG}_cnn_pmc_sq.err || { rm -rf $OUT/${TAG}_cnn_pmc_sq; rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/${TAG}_cnn_pmc_sq -o run --output-format csv -- $CNN > /dev/null 2>> $OUT/${TAG}_cnn_pmc_sq.err; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_cnn_pmc_fetch -o run --output-format csv -- $CNN > /dev/null 2> $OUT/${TAG}_cnn_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_cnn_pmc_write -o run --output-format csv -- $CNN > /dev/null 2> $OUT/${TAG}_cnn_pmc_write.err
# 5. the int16-native path (raw ADC samples in HBM): the same traffic pass, to set beside the float32 one
I16="python3 bench.py --int16 --steps 2 --warmup 1 --no-secondary --cpu-sample 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_i16_pmc_fetch -o run --output-format csv -- $I16 > /dev/null 2> $OUT/${TAG}_i16_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_i16_pmc_write -o run --output-format csv -- $I16 > /dev/null 2> $OUT/${TAG}_i16_pmc_write.err
# 5b. Pareto lengths and the preset's default window: the same traffic passes (their roofline objects carried traffic: null until round 5)
PAR="python3 bench.py --lens pareto --steps 2 --warmup 1 --no-secondary --cpu-sample 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_par_pmc_fetch -o run --output-format csv -- $PAR > /dev/null 2> $OUT/${TAG}_par_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_par_pmc_write -o run --output-format csv -- $PAR > /dev/null 2> $OUT/${TAG}_par_pmc_write.err
DEF="python3 bench.py --max_obs_trace 16000 --steps 2 --warmup 1 --no-secondary --cpu-sample 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_def_pmc_fetch -o run --output-format csv -- $DEF > /dev/null 2> $OUT/${TAG}_def_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_def_pmc_write -o run --output-format csv -- $DEF > /dev/null 2> $OUT/${TAG}_def_pmc_write.err
# 6. summaries (the raw traces stay on the box: only these come back)
S=$OUT/${TAG}_summary
mkdir -p $S
python3 profiles/summarize_phases.py $OUT/${TAG}_stats/run_results.db $S/kernel_stats_by_workload.csv > $S/kernel_stats_by_workload.txt
python3 profiles/summarize_pmc.py $OUT/${TAG}_pmc_fetch/run_counter_collection.csv $OUT/${TAG}_pmc_write/run_counter_collection.csv 96000 200000 $S/traffic.json > $S/traffic.txt
python3 profiles/summarize_pmc.py $OUT/${TAG}_i16_pmc_fetch/run_counter_collection.csv $OUT/${TAG}_i16_pmc_write/run_counter_collection.csv 96000 200000 $S/traffic_int16.json > $S/traffic_int16.txt
python3 profiles/summarize_pmc.py $OUT/${TAG}_par_pmc_fetch/run_counter_collection.csv $OUT/${TAG}_par_pmc_write/run_counter_collection.csv 96000 200000 $S/traffic_pareto.json > $S/traffic_pareto.txt
python3 profiles/summarize_pmc.py $OUT/${TAG}_def_pmc_fetch/run_counter_collection.csv $OUT/${TAG}_def_pmc_write/run_counter_collection.csv 96000 16000 $S/traffic_default_window.json > $S/traffic_default_window.txt
python3 profiles/summarize_pmc.py --stats $OUT/${TAG}_cnn_stats/run_results.db $S/cnn200k_kernel_stats.csv > $S/cnn200k_kernel_stats.txt
python3 profiles/summarize_pmc.py $OUT/${TAG}_cnn_pmc_fetch/run_counter_collection.csv $OUT/${TAG}_cnn_pmc_write/run_counter_collection.csv 8000 200000 $S/cnn200k_traffic.json > $S/cnn200k_traffic.txt
python3 profiles/summarize_sq.py $OUT/${TAG}_cnn_pmc_sq/run_counter_collection.csv $S/sq_counters_cnn200k.json > $S/sq_counters_cnn200k.txt
cp $OUT/${TAG}_bench.json $OUT/${TAG}_bench_under_rocprof.json $OUT/${TAG}_cnn_under_rocprof.json $S/
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_cnn_stats $OUT/${TAG}_cnn_pmc_sq $OUT/${TAG}_cnn_pmc_fetch $OUT/${TAG}_cnn_pmc_write $OUT/${TAG}_i16_pmc_fetch $OUT/${TAG}_i16_pmc_write $OUT/${TAG}_par_pmc_fetch $OUT/${TAG}_par_pmc_write $OUT/${TAG}_def_pmc_fetch $OUT/${TAG}_def_pmc_write
ls $S
