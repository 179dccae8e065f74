#!/bin/bash
# round 2, second GPU call: the whole GPU suite after the transition-header refactor (reset counter, SafeInterruptibility,
# switches), and where the host clock goes in a 20-step bench.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -30 $O/pytest_gpu.log
for i in 1 2; do SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2> $O/trace_20_$i.err | cut -c1-300; tail -2 $O/trace_20_$i.err; done
SGK_NO_GRAPH=1 SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fused 2> $O/trace_20_eager.err | cut -c1-300; tail -2 $O/trace_20_eager.err
SGK_NO_GRAPH=1 SGK_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-fused 2> $O/trace_2000_eager.err | cut -c1-300; tail -2 $O/trace_2000_eager.err
