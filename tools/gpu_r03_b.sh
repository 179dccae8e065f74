#!/bin/bash
# round 3, second GPU call: what separates the streamed rollout's ring form (5.5 us per step) from the pure store pattern
# (4.9 us, tools/exp_write_patterns.hip): resident waves per CU (__launch_bounds__ builds) x grid size (rounds / tail), and the
# waves' drift (the 100 steps of a pass as several shorter launches)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
export SGK_NO_BUILD=1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe > $O/wp_build.log 2>&1
{ timeout 300 /tmp/wp_probe "chunked"; timeout 300 /tmp/wp_probe "ring size fine"; timeout 300 /tmp/wp_probe "base slice sc1"; } > $O/write_patterns_chunked.log 2>&1
tail -3 $O/write_patterns_chunked.log
L=$PWD/safe-grid-agents_amd/lib
{
for lib in libsgk.so libsgk_w6.so libsgk_w8.so; do
  for g in 1024 1280 1536 2048 2560 3072 4096; do
    SGK_LIB_PATH=$L/$lib SGK_STREAM_GRID=$g timeout 300 python tools/exp_stream_grid.py 2>&1 | grep "n="
  done
done
} > $O/stream_grid_x_occupancy.log 2>&1
cat $O/stream_grid_x_occupancy.log
{
SGK_STREAM_GRID=4096 timeout 300 python tools/exp_stream_grid.py --rings 100 --chunks 100,50,25,20,10,5 2>&1 | grep "n="
SGK_STREAM_GRID=1024 timeout 300 python tools/exp_stream_grid.py --rings 100 --chunks 100,50,25,20,10,5 2>&1 | grep "n="
SGK_STREAM_GRID=1024 timeout 300 python tools/exp_stream_grid.py --rings 100 --chunks 100,25,10 --layout tile 2>&1 | grep "n="
} > $O/stream_chunked_launches.log 2>&1
cat $O/stream_chunked_launches.log
python tools/write_bw_probe.py > $O/write_bw_probe.log 2>&1; cat $O/write_bw_probe.log
