#!/bin/bash
# round 3, call v: whole rounds (residency lowered by LDS padding) in the streamed rollout: forced 2..5 and auto, in place and ring
cd "$GRAFT_REPO_ROOT" || exit 1
export SGK_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/v; mkdir -p $O
for r in -1 5 4 3 2 0; do
  echo "== SGK_STREAM_RESIDENT=$r" >> $O/stream_resident.log
  SGK_STREAM_RESIDENT=$r timeout 300 python tools/bench_stream.py --envs BoatRace-v0 --sizes 524288,1048576 --k 100 --ring 100 --modes stream --reps 10 2>&1 | grep "n=" >> $O/stream_resident.log
done
SGK_STREAM_RESIDENT=-1 timeout 300 python tools/bench_stream.py --envs IslandNavigation-v0,SideEffectsSokoban-v0,TomatoWatering-v0 --sizes 1048576 --k 100 --ring 100 --modes stream --reps 5 2>&1 | grep "n=" >> $O/stream_resident.log
echo "== auto" >> $O/stream_resident.log
timeout 300 python tools/bench_stream.py --envs IslandNavigation-v0,SideEffectsSokoban-v0,TomatoWatering-v0 --sizes 1048576 --k 100 --ring 100 --modes stream --reps 5 2>&1 | grep "n=" >> $O/stream_resident.log
cat $O/stream_resident.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -x -q -k "stream or ring or tile" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
