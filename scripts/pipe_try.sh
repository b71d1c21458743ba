run() { echo "== $*"; env "$@" CACHE=128000000 BLOCKS=6 PRE=3 Q=256 E=100 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | grep block | awk '{g+=$3; s+=$5; e+=$7; h+=$10; n++} END {printf "avg %.0f games/s %.2f Msims/s %.2f Mevals/s hit %.3f\n", g/n, s/n, e/n, h/n}'; }
run AZMI_PIPE_MIN_ACTIVE=0 AZMI_PIPE_INLINE=4 &&
run AZMI_PIPE_MIN_ACTIVE=3 AZMI_PIPE_INLINE=5 &&
run AZMI_PIPE_MIN_ACTIVE=4 AZMI_PIPE_INLINE=5 &&
run AZMI_PIPE_MIN_ACTIVE=3 AZMI_PIPE_INLINE=6 &&
run AZMI_PIPE_MIN_ACTIVE=5 AZMI_PIPE_INLINE=6 &&
run AZMI_PIPE_MIN_ACTIVE=0 AZMI_PIPE_INLINE=4
