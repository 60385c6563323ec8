#!/bin/bash
# Profiling evidence of a round, collected on the GPU box (through gpurun, from the repo root):  bash tools/collect_profiles.sh r03
# Outputs under gpurun_out/<tag>/ (the summaries are then copied into profiles/ as <tag>_*):
#   bench_line.json        the default bench line (sampler in the step, all legs, rocprofv3 --pmc traffic passes, CPU baseline)
#   bench_line_serial.json one stream
#   step_launches.txt      every launch of one C2 step (hipEvent brackets): us, GFLOP, TF/s, MB, GB/s
#   kernel_stats.txt       rocprofv3 --kernel-trace --stats of the serial C2 bench (the durations roofline.avg_launch_us must agree with)
#   pmc_sq.txt             rocprofv3 --pmc SQ_* (separate pass) of the same command
#   c3_*, c5_*             the same for the C3 probe (bench.py --probe c3) and the C5 probe (bench.py --probe c5)
#   gemm_sweep.txt, slowfast_layers.txt, attention / bias-build probes
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
C2="python bench.py --streams 1 --legs c2 --no-cpu-baseline --no-pmc"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err
timeout 300 $C2 --steps 20 --warmup 5 > $out/bench_line_serial.json 2>> $out/bench.err
timeout 300 python tools/profile_step.py > $out/step_launches.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $C2 --steps 20 --warmup 3 --min-timed-s 0 > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > $out/kernel_stats.txt
timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $out/pmc_sq -o s -- $C2 --steps 2 --warmup 1 --profile-steps 0 --min-timed-s 0 > $out/pmc_s.log 2>&1
f=$(find $out/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_sq_summary.py $f $tag > $out/pmc_sq.txt
for leg in c3 c5; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/${leg}_trace -o t -- python bench.py --probe $leg --probe-steps 3 > $out/${leg}_trace.log 2>&1
  db=$(find $out/${leg}_trace -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > $out/${leg}_kernel_stats.txt
  timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $out/${leg}_pmc_sq -o s -- python bench.py --probe $leg --probe-steps 1 > $out/${leg}_pmc_s.log 2>&1
  f=$(find $out/${leg}_pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_sq_summary.py $f $tag > $out/${leg}_pmc_sq.txt
done
timeout 300 python tools/sf_layers.py > $out/slowfast_layers.txt 2>&1
timeout 300 python tools/gemm_sweep.py > $out/gemm_sweep.txt 2>&1
timeout 300 python tools/swinb_probe.py 4 table > $out/c5_launches.txt 2>&1
timeout 300 python tools/bias_build_probe.py > $out/bias_build.txt 2>&1
find $out -name "*.db" -size +20M -delete
find $out -name "*counter_collection.csv" -size +20M -delete
ls $out
