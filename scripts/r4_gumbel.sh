#!/bin/bash
# Connect4 4096 x 800 with Gumbel roots: lock-step rounds (4 shards) against the pipeline's generic tree kernel
cd "$(dirname "$0")/.."
timeout -k 10 400 python bench.py --gumbel --driver rounds --steps 10 --warmup 3 --no-cpu-baseline --no-tawlbwrdd > gpurun_out/r4_gumbel_rounds.json 2> gpurun_out/r4_gumbel_rounds.err
echo "rounds rc=$?" > gpurun_out/r4_gumbel.txt
timeout -k 10 400 python bench.py --gumbel --driver pipeline --steps 10 --warmup 3 --no-cpu-baseline --no-tawlbwrdd > gpurun_out/r4_gumbel_pipe.json 2> gpurun_out/r4_gumbel_pipe.err
echo "pipe rc=$?" >> gpurun_out/r4_gumbel.txt
