#!/bin/bash
# A/B of two builds of libazmi.so on one box: alphazero-pybind11_amd/libazmi_A.so (baseline) against libazmi.so
cd "$(dirname "$0")/.."
out=gpurun_out/big_ab.txt; : > $out
for i in 1 2; do
  AZMI_LIB=$PWD/alphazero-pybind11_amd/libazmi_A.so timeout -k 10 200 python scripts/big_ab.py >> $out 2>&1 || exit 1
  timeout -k 10 200 python scripts/big_ab.py >> $out 2>&1 || exit 1
done
