#!/bin/bash
# SQ / TCC counters of one attention geometry (separate --pmc passes; no trace domains).  usage: attn32_pmc.sh bin "128 3 4 392 64" tag [kernel name part]
bin=$1; geo=$2; tag=$3; kname=${4:-window_attention32}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/attn_pmc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU"
i=0
for P in "$P1" "$P2" "$P3" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $out/p$i -o c -- $root/tools/ubench/$bin $geo 3 > $out/p$i.log 2>&1
done
python $root/tools/pmc_kernel.py $kname $(find $out -name "*counter_collection.csv") > $root/gpurun_out/attn_pmc_$tag.txt
cat $root/gpurun_out/attn_pmc_$tag.txt
