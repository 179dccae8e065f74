#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run() {
  make -s -C safe-grid-agents_amd/csrc OUT=../lib/libsgk_x.so OBJDIR=build_x EXTRA="$2" -j8 > $O/build_$1.log 2>&1
  echo "== $1 ($2)"; SGK_LIB_PATH=$PWD/safe-grid-agents_amd/lib/libsgk_x.so timeout 300 python tools/prof_tabq_kernels.py 262144 2>&1 | grep "act"
  rm -rf safe-grid-agents_amd/lib/libsgk_x.so safe-grid-agents_amd/csrc/build_x
}
{ run base "-DSGK_DBG_ACT=0"; run one_f64_compare "-DSGK_DBG_ACT=64"; run action_from_si "-DSGK_DBG_ACT=128"; run action_from_bits "-DSGK_DBG_ACT=256"; } > $O/act_experiment3.log 2>&1
cat $O/act_experiment3.log
