#!/bin/bash
# round 3, call n: the ring written by 1024-lane workgroups (16 tiles = 25.6 KB contiguous per step), one or two per CU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/n; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe 2>/dev/null
WP_INDEX=0 timeout 60 /tmp/wp_probe "" > $O/ring_big.log 2>&1
timeout 300 /tmp/wp_probe "1024-lane" >> $O/ring_big.log 2>&1
WP_INDEX=0 timeout 60 /tmp/wp_probe "" >> $O/ring_big.log 2>&1
grep -v "^fill\|^#" $O/ring_big.log
