#!/bin/bash
# round 2, experiment: cache policy of the streamed rollout's stores (board tiles: sc1 / nt / sc1+nt / plain; records: plain / sc1)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run() { # name, EXTRA
  make -s -C safe-grid-agents_amd/csrc OUT=../lib/libsgk_x.so OBJDIR=build_x EXTRA="$2" -j8 > $O/build_$1.log 2>&1
  echo "== $1 ($2)"
  for i in 1 2 3; do SGK_STREAM_GRID=4096 SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_x.so timeout 600 python tools/bench_stream.py --envs BoatRace-v0 --sizes 1048576 --ring 100 --modes stream --reps 10 2>&1 | grep n=; done
  rm -rf safe-grid-agents_amd/lib/libsgk_x.so safe-grid-agents_amd/csrc/build_x
}
{
run sc1 "-DSGK_BOARD_STORE_AUX=16"
run plain "-DSGK_BOARD_STORE_AUX=0"
run sc0sc1 "-DSGK_BOARD_STORE_AUX=17"
run sc1_recsc1 "-DSGK_BOARD_STORE_AUX=16 -DSGK_STREAM_REC_SC1=1"
run plain_recsc1 "-DSGK_BOARD_STORE_AUX=0 -DSGK_STREAM_REC_SC1=1"
run sc0sc1_recsc1 "-DSGK_BOARD_STORE_AUX=17 -DSGK_STREAM_REC_SC1=1"
} > $O/store_policy_repeats.log 2>&1
cat $O/store_policy_repeats.log
