# round 6: the Connect4 tile's experiment builds (scripts/micro/tile_exp), one binary per knob set, all on one box
# usage: KNOBS="-DX_M0=1|-DX_SCHED=1|..." bash scripts/gpu_tile_exp.sh   (| separates knob sets; the empty set = the product tile)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/scripts/micro/tile_exp
out=$GRAFT_REPO_ROOT/gpurun_out/r6_tile_exp.txt; : > $out
IFS='|' read -ra SETS <<< "${KNOBS:-}"
[ ${#SETS[@]} -eq 0 ] && SETS=("")
i=0
for k in "base" "${SETS[@]}"; do
  [ "$k" = "base" ] && k=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $FLAGS $k tile_exp.hip -o /tmp/tx_$i || { echo "build failed: $k" >> $out; i=$((i+1)); continue; }
  i=$((i+1))
done
# two passes over the binaries (box warm-up / clock drift shows as a difference between the passes)
for pass in 1 2; do
  j=0
  for k in "base" "${SETS[@]}"; do
    [ -x /tmp/tx_$j ] && { echo "== pass $pass [$k]" >> $out; timeout -k 5 120 /tmp/tx_$j >> $out 2>&1; }
    j=$((j+1))
  done
done
cat $out
