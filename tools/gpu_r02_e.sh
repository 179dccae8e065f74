#!/bin/bash
# round 2, fifth GPU call: LDS-image tile writer: GPU suite, stream timings (A/B vs the bpermute build), bench at both flag sets
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log | cut -c1-300
timeout 900 python tools/bench_stream.py --envs BoatRace-v0,IslandNavigation-v0,SideEffectsSokoban-v0,AbsentSupervisor-v0 --ring 100 > $O/bench_stream_lds.log 2>&1; cat $O/bench_stream_lds.log
# A/B: the same binary built with the register / ds_bpermute tile writer
make -s -C safe-grid-agents_amd/csrc OUT=../lib/libsgk_bperm.so OBJDIR=build_bperm EXTRA=-DSGK_TILE_IN_LDS=0 -j8 > $O/build_bperm.log 2>&1
SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_bperm.so timeout 900 python tools/bench_stream.py --envs BoatRace-v0,SideEffectsSokoban-v0 --sizes 131072,1048576 --ring 100 > $O/bench_stream_bperm.log 2>&1; cat $O/bench_stream_bperm.log
rm -rf safe-grid-agents_amd/lib/libsgk_bperm.so safe-grid-agents_amd/csrc/build_bperm
for g in 2048 4096; do echo "SGK_STREAM_GRID=$g"; SGK_STREAM_GRID=$g timeout 300 python tools/bench_stream.py --envs BoatRace-v0 --sizes 131072,1048576 --ring 100 2>&1 | grep n=; done > $O/stream_grid_sweep.log 2>&1; cat $O/stream_grid_sweep.log
for i in 1 2; do SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_$i.log 2> $O/bench_20_$i.err; tail -1 $O/bench_20_$i.log | cut -c1-330; grep "bench trace" $O/bench_20_$i.err | head -1; done
timeout 900 python bench.py --gpus 1 --steps 2000 --warmup 200 > $O/bench_2000.log 2>&1; tail -1 $O/bench_2000.log | cut -c1-1800
timeout 900 python tools/bench_configs.py > $O/configs.log 2>&1; cat $O/configs.log | cut -c1-400
