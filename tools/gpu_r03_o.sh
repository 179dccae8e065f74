#!/bin/bash
# round 3, call o: does the tile piece's alignment matter? 1600-byte pieces put every other tile boundary in the middle of a 128-byte line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/o; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_write_patterns.hip -o /tmp/wp_probe 2>/dev/null
for piece in 1600 1536 1664 1792 2048 1600 3072 3200 4096; do
  echo "== piece $piece bytes" >> $O/ring_piece_alignment.log
  WP_INDEX=0 timeout 60 /tmp/wp_probe "" $piece 2>&1 | grep -v "^fill\|^#\|^variant" >> $O/ring_piece_alignment.log
  WP_RING=32 timeout 60 /tmp/wp_probe "slice ring size" $piece 2>&1 | grep -v "^fill\|^#\|^variant" >> $O/ring_piece_alignment.log
done
cat $O/ring_piece_alignment.log
