# the weight-stationary convolution micro-benchmark (scripts/micro/conv_stationary.hip) on the GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w scripts/micro/conv_stationary.hip -o /tmp/cs || exit 1
timeout -k 10 120 /tmp/cs > gpurun_out/r5_conv_stationary.txt 2>&1 || { cat gpurun_out/r5_conv_stationary.txt; exit 1; }
cat gpurun_out/r5_conv_stationary.txt
