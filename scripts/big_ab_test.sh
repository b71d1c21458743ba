#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "tawlbwrdd or tafl or brandubh or opentafl or gumbel or stargambit or wide or big or rules or replay or leafnet or net or t3 or spatial" > gpurun_out/r4_big_tests.log 2>&1; echo rc=$? >> gpurun_out/r4_big_tests.log
tail -3 gpurun_out/r4_big_tests.log
grep -q "rc=0" gpurun_out/r4_big_tests.log || exit 1
bash scripts/big_ab.sh
