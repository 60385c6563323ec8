#!/bin/bash
# Round-2 profiling evidence, collected on the GPU box (through gpurun, from the repo root):  bash tools/collect_profiles_r02.sh r02a
# Outputs under gpurun_out/<tag>/ (the summaries are then copied into profiles/):
#   C2: the default bench line (sampler in the step + all legs), the serial line, the per-launch step table, rocprofv3
#       --kernel-trace --stats of the serial bench, separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters)
#   C3: kernel stats of one video through both branches (tools/c3_probe.py), SQ counters of the SlowFast branch
#   C5: kernel stats + SQ counters of Swin-B on 64x256x256 clips (tools/swinb_probe.py)
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
C2="python bench.py --streams 1 --legs c2 --no-cpu-baseline"
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err
timeout 300 $C2 --steps 20 --warmup 5 > $out/bench_line_serial.json 2>> $out/bench.err
timeout 300 python tools/profile_step.py > $out/step_launches.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $C2 --steps 20 --warmup 3 > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > $out/kernel_stats.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- $C2 --steps 2 --warmup 1 --profile-steps 0 > $out/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- $C2 --steps 2 --warmup 1 --profile-steps 0 > $out/pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -o s -- $C2 --steps 2 --warmup 1 --profile-steps 0 > $out/pmc_s.log 2>&1
# C3
timeout 300 python tools/c3_probe.py > $out/c3_probe.log 2>&1
timeout 300 python tools/sf_probe.py >> $out/c3_probe.log 2>&1
timeout 300 python tools/sf_layers.py > $out/slowfast_layers.txt 2>&1
KVQ_SF_FUSE_FAST=0 KVQ_CONV_TAP_TABLE=1 timeout 300 python tools/sf_layers.py > $out/slowfast_layers_conv_by_conv.txt 2>&1
timeout 300 python tools/gemm_sweep.py > $out/gemm_sweep.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/c3_trace -o t -- python tools/c3_probe.py > $out/c3_trace.log 2>&1
db=$(find $out/c3_trace -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > $out/c3_kernel_stats.txt
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/c3_pmc_sq -o s -- python tools/sf_probe.py 2 > $out/c3_pmc_s.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c3_pmc_fetch -o f -- python tools/sf_probe.py 2 > $out/c3_pmc_f.log 2>&1
# C5
timeout 300 python tools/swinb_probe.py 4 1 > $out/c5_probe.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/c5_trace -o t -- python tools/swinb_probe.py 4 > $out/c5_trace.log 2>&1
db=$(find $out/c5_trace -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > $out/c5_kernel_stats.txt
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/c5_pmc_sq -o s -- python tools/swinb_probe.py 2 > $out/c5_pmc_s.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c5_pmc_fetch -o f -- python tools/swinb_probe.py 2 > $out/c5_pmc_f.log 2>&1
find $out -name "*.db" -size +20M -delete
ls $out
