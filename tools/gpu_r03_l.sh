#!/bin/bash
# round 3, call l: the ring written in ONE-SHOT shape (a workgroup per (step, four tiles), step-major) against the persistent shape
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/l; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe 2>/dev/null
WP_INDEX=0 timeout 60 /tmp/wp_probe "" > $O/ring_oneshot.log 2>&1
timeout 300 /tmp/wp_probe "ONE-SHOT" >> $O/ring_oneshot.log 2>&1
WP_INDEX=0 timeout 60 /tmp/wp_probe "" >> $O/ring_oneshot.log 2>&1
grep -v "^fill\|^#" $O/ring_oneshot.log
