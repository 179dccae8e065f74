#!/bin/bash
# tabular-Q on the GPU box: parity tests, the fused rollout per env / agent count (AUTO, forced LDS-resident, forced HBM-resident)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r04_tabq}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 -k "tabq" > $O/pytest_tabq.log 2>&1; tail -2 $O/pytest_tabq.log
for k in auto lds hbm; do
  SGK_BENCH_KERNEL=$k SGK_BENCH_SIZES=16384,65536,131072,262144,1048576 timeout 600 python tools/bench_tabq_sizes.py IslandNavigation-v0 BoatRace-v0 DistributionalShift-v0 2>&1 | grep -v amdgpu.ids > $O/bench_tabq_sizes_$k.log; cat $O/bench_tabq_sizes_$k.log
done
