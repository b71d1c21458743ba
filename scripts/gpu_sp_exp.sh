# spatial tile timing variants (scripts/micro/tile_exp/sp_exp.hip), one binary per early-exit point
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/scripts/micro/tile_exp && python3 make_spx.py
out=$GRAFT_REPO_ROOT/gpurun_out/r6_sp_tile_timing.txt; : > $out
for d in 1 2 3 4 5 0; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DX_DBG=$d $EXTRA sp_exp.hip -o /tmp/spx_$d || exit 1; done
for d in 1 2 3 4 5 0; do timeout -k 5 120 /tmp/spx_$d >> $out 2>&1; done
sort -k1,1 -k5,5n -k3,3n $out
