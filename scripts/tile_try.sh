cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/c4_tile_timing.hip -o /tmp/c4t || exit 1
C4T_NO_CLOCK=1 /tmp/c4t 714 2904 2>&1 | grep -E "rows|full|no weight DMA|no fragment|no MFMA|stem"
timeout -k 10 300 python -m pytest tests/test_gpu_leafnet.py -x -q -k "connect4 or c4 or fixture or big_and_small or batch_shapes or bf16x3" 2>&1 | tail -3
