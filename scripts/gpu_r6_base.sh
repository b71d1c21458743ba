# round 6 baseline on one box: the tile's timing variants, the tile inside the pipeline (AZMI_PIPE_PROF), the 4096 x 800 pipeline rate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/c4_tile_timing.hip -o /tmp/c4t || exit 1
C4T_NO_CLOCK=1 timeout -k 10 200 /tmp/c4t 714 2904 > gpurun_out/r6_tile_timing_base.txt 2>&1 || exit 1
cat gpurun_out/r6_tile_timing_base.txt
bash scripts/tile_in_mix.sh && mv gpurun_out/r5_tile_in_mix.txt gpurun_out/r6_tile_in_mix_base.txt && tail -25 gpurun_out/r6_tile_in_mix_base.txt
CACHE=128000000 Q=256 E=80 BLOCKS=4 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py > gpurun_out/r6_pipe_base.txt 2>&1; tail -6 gpurun_out/r6_pipe_base.txt
