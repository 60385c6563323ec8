#!/bin/bash
# SQ / L2 counters of one kernel (separate --pmc passes; no trace domains).  usage: kernel_pmc.sh tag kernel-name-filter command...
tag=$1; flt=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/kpmc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$P1" "$P2" "$P3" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd $root && timeout 300 rocprofv3 --pmc $P --output-format csv -d $out/p$i -o c -- "$@" > $out/p$i.log 2>&1)
done
python $root/tools/pmc_kernel.py "$flt" $(find $out -name "*counter_collection.csv") > $root/gpurun_out/kpmc_$tag.txt
cat $root/gpurun_out/kpmc_$tag.txt
