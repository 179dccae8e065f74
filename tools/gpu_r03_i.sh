#!/bin/bash
# round 3, call i: the leaner streamed loop (hoisted destinations, one LDS round trip per flush): GPU suite, small-batch and
# 1 M rates of every form, the 2048-tile store probe for DESIGN 3.2
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 600 python tools/bench_stream.py --envs BoatRace-v0 --sizes 65536,131072,262144,1048576 --k 100 --ring 100 --modes stream --reps 10 > $O/stream_sizes.log 2>&1
grep "n=" $O/stream_sizes.log
timeout 600 python tools/bench_stream.py --envs IslandNavigation-v0,SideEffectsSokoban-v0,TomatoWatering-v0 --sizes 65536,1048576 --k 100 --ring 100 --modes stream --reps 5 > $O/stream_envs.log 2>&1
grep "n=" $O/stream_envs.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe 2>/dev/null
for pat in "base slice sc1" "slice ring size" "slice sc1 delay" "tile-major burst sc1"; do /tmp/wp_probe "$pat" 1600 2048; done > $O/write_patterns_2048_tiles.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; tail -c 600 $O/bench_20.json
