#!/bin/bash
# SQ / TCC counters of one pgemm_probe case: tools/probes/pgemm_pmc.sh <tag> M N K cfg atr btr splits
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format rocpd -d $out -o a -- $GRAFT_REPO_ROOT/tools/probes/pgemm_probe one "$@" 3 > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU \
  --output-format rocpd -d $out -o b -- $GRAFT_REPO_ROOT/tools/probes/pgemm_probe one "$@" 3 > $out/b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE \
  --output-format rocpd -d $out -o c -- $GRAFT_REPO_ROOT/tools/probes/pgemm_probe one "$@" 3 > $out/c.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $out -name '*.db'); do echo "#### $f"; python tools/rocpd_pmc.py $f gemm_kernel; done > $out/summary.txt 2>&1
cat $out/a.log | tail -2
cat $out/summary.txt
