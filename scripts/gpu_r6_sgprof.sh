cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_tafl_family.py tests/test_gpu_parity_tafl.py -x -q 2>&1 | tail -3
AZMI_LIB=$GRAFT_REPO_ROOT/alphazero-pybind11_amd/libazmi_prof.so GAME=stargambit timeout -k 10 300 python scripts/big_prof.py > gpurun_out/r6_sg_phase_prof.txt 2>&1; cat gpurun_out/r6_sg_phase_prof.txt | tail -20
