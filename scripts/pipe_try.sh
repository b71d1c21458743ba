cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BLOCKS=1 PRE=0.5 E=20 timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pipe -o pipe -- python scripts/pipe_bench.py > gpurun_out/prof_pipe.log 2>&1
find gpurun_out/prof_pipe -name "*stats*" | head -5
f=$(find gpurun_out/prof_pipe -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-260
