cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/c4_tile_timing.hip -o /tmp/c4t || exit 1
C4T_NO_CLOCK=1 /tmp/c4t 714 2>&1 | grep -E "rows|full|stem|\.\.\."
timeout -k 10 400 python -m pytest tests/test_gpu_leafnet.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
