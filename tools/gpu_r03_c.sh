#!/bin/bash
# round 3, third GPU call: the XCD-contiguous tile assignment (probe + streamed rollout kernel A/B) and the GPU suite
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
export SGK_NO_BUILD=1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe > $O/wp_build.log 2>&1
{ timeout 300 /tmp/wp_probe "XCD"; timeout 300 /tmp/wp_probe "base slice sc1";  timeout 300 /tmp/wp_probe "tile-major sc1"; timeout 300 /tmp/wp_probe "slice ring size"; } > $O/write_patterns_xcd.log 2>&1
cat $O/write_patterns_xcd.log | grep -v "^fill"
{
for x in 0 1; do
  for g in 1024 1280 2048 4096; do
    SGK_STREAM_XCD=$x SGK_STREAM_GRID=$g timeout 300 python tools/exp_stream_grid.py --rings 32,64,100 2>&1 | grep "n=" | sed "s/^/xcd=$x /"
  done
done
for x in 0 1; do
  SGK_STREAM_XCD=$x timeout 300 python tools/exp_stream_grid.py --rings 32,100 --layout tile 2>&1 | grep "n=" | sed "s/^/tile-major xcd=$x /"
  SGK_STREAM_XCD=$x timeout 300 python tools/exp_stream_grid.py --env IslandNavigation-v0 --rings 32,100 2>&1 | grep "n=" | sed "s/^/xcd=$x /"
  SGK_STREAM_XCD=$x timeout 300 python tools/exp_stream_grid.py --env SideEffectsSokoban-v0 --rings 32,100 2>&1 | grep "n=" | sed "s/^/xcd=$x /"
  SGK_STREAM_XCD=$x timeout 300 python tools/exp_stream_grid.py --n 131072 --rings 100 2>&1 | grep "n=" | sed "s/^/xcd=$x /"
done
} > $O/stream_xcd_ab.log 2>&1
cat $O/stream_xcd_ab.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
