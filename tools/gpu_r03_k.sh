#!/bin/bash
# round 3, call k: does bounding a wave's outstanding stores tighten the write front? fill shapes + the ring probe with vmcnt caps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/k; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_fill_shapes.hip -o /tmp/fill_shapes 2>/dev/null && timeout 300 /tmp/fill_shapes > $O/fill_shapes.log 2>&1
cat $O/fill_shapes.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe 2>/dev/null
WP_INDEX=0 /tmp/wp_probe "" > $O/ring_capped.log 2>&1
timeout 600 /tmp/wp_probe "stores in flight capped" >> $O/ring_capped.log 2>&1
grep -v "^fill\|^#" $O/ring_capped.log
