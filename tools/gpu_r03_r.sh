#!/bin/bash
# round 3, call r: ring slices are 25 MiB (boards) and 4 MiB (records) apart -- every slice's tile w on the same channel and bank?
# the probe with the slice strides padded
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe 2>/dev/null
for pad in "0,0" "4096,4096" "65536,65536" "69632,4352" "1048576,1048576" "1114112,69632" "2162688,266240" "0,69632" "69632,0" "0,0"; do
  echo "== WP_SLICE_PAD=$pad" >> $O/ring_slice_pad.log
  WP_SLICE_PAD=$pad WP_INDEX=0 timeout 60 /tmp/wp_probe "" 2>&1 | grep -v "^fill\|^#\|^variant" >> $O/ring_slice_pad.log
  WP_SLICE_PAD=$pad WP_RING=32 timeout 60 /tmp/wp_probe "slice ring size" 2>&1 | grep -v "^fill\|^#\|^variant" >> $O/ring_slice_pad.log
done
cat $O/ring_slice_pad.log
