#!/bin/bash
# quick A/B over SGK_PARTITIONS
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for P in 1 2 3 4; do
  echo "== SGK_PARTITIONS=$P"
  SGK_PARTITIONS=$P python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  SGK_PARTITIONS=$P python tools/sweep.py BoatRace-v0 65536 262144 1048576 4194304 2>&1 | grep compact | cut -c1-120
done
SGK_PARTITIONS=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_rollouts or sharding or million" 2>&1 | tail -3
