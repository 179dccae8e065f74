#!/bin/bash
# round 3, call t: what do the write-path counters say about FAST and SLOW allocations of the ring? (per dispatch, with durations)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/t; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp_ring_alloc_lottery.hip -o /tmp/lottery 2>/dev/null
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
           "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- /tmp/lottery pmc > $O/p$i.log 2>&1
  for f in $(find $O/p$i -name "*counter_collection.csv"); do cp $f $O/counters_$i.csv; done
  for f in $(find $O/p$i -name "*kernel_trace.csv"); do cp $f $O/trace_$i.csv; done
  rm -rf $O/p$i
done
ls -la $O; head -3 $O/counters_1.csv; head -3 $O/trace_1.csv; tail -3 $O/p1.log
