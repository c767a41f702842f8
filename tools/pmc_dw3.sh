#!/bin/bash
# SQ / TCC counters of the bf16-pipe dW kernel (microbench at B=$1): gpurun_out/pmc_dw3/
B=${1:-500}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_dw3
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format rocpd -d $out -o a -- python tools/microbench.py $B > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM \
  --output-format rocpd -d $out -o b -- python tools/microbench.py $B > $out/b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE \
  --output-format rocpd -d $out -o c -- python tools/microbench.py $B > $out/c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum \
  --output-format rocpd -d $out -o d -- python tools/microbench.py $B > $out/d.log 2>&1
for f in $(find $out -name '*.db'); do echo "#### $f"; python tools/rocpd_pmc.py $f dw3_kernel; done > $out/summary.txt 2>&1
tail -2 $out/c.log
cat $out/summary.txt
