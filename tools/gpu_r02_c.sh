#!/bin/bash
# round 2, third GPU call: GPU suite (wave-private tiles, streamed rollout, new env) + first timings of the streamed rollout
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -40 $O/pytest_gpu.log | cut -c1-400
timeout 900 python tools/bench_stream.py --envs BoatRace-v0,IslandNavigation-v0,SideEffectsSokoban-v0 --ring 100 > $O/bench_stream.log 2>&1; cat $O/bench_stream.log
timeout 600 python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_2000.log 2>&1; tail -1 $O/bench_2000.log | cut -c1-1500
