#!/bin/bash
# The round's differential soak (GPU box): every mode of tests/soak_vs_oracle.py, outputs under gpurun_out/<tag>_soak_*.txt
TAG=${1:-r02}
X=${2:-1}   # multiplier of the rounds per mode
O=gpurun_out
set -e
python tests/soak_vs_oracle.py $((60 * X)) > $O/${TAG}_soak_llr.txt 2>&1
python tests/soak_vs_oracle.py $((30 * X)) big > $O/${TAG}_soak_big.txt 2>&1
python tests/soak_vs_oracle.py $((60 * X)) candidates > $O/${TAG}_soak_cand.txt 2>&1
python tests/soak_vs_oracle.py $((60 * X)) start_peak > $O/${TAG}_soak_sp.txt 2>&1
python tests/soak_vs_oracle.py $((80 * X)) cnn > $O/${TAG}_soak_cnn.txt 2>&1
python tests/soak_vs_oracle.py $((300 * X)) predict > $O/${TAG}_soak_predict.txt 2>&1
python tests/soak_vs_oracle.py $((60 * X)) int16 > $O/${TAG}_soak_int16.txt 2>&1
tail -n 1 $O/${TAG}_soak_*.txt
