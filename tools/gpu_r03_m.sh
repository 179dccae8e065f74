#!/bin/bash
# round 3, call m: do a workgroup's waves writing TOGETHER (a barrier per iteration / step) get the one-shot rate?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_fill_shapes.hip -o /tmp/fill_shapes 2>/dev/null && timeout 300 /tmp/fill_shapes > $O/fill_shapes.log 2>&1
grep -i "barrier\|one workgroup per 4\|grid-stride, " $O/fill_shapes.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe 2>/dev/null
WP_INDEX=0 timeout 60 /tmp/wp_probe "" > $O/ring_sync.log 2>&1
timeout 300 /tmp/wp_probe "workgroup barrier" >> $O/ring_sync.log 2>&1
WP_INDEX=0 timeout 60 /tmp/wp_probe "" >> $O/ring_sync.log 2>&1
grep -v "^fill\|^#" $O/ring_sync.log
