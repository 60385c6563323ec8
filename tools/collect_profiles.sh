#!/bin/bash
# Collects the round's profiling evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r01n
# Outputs under gpurun_out/<tag>/ (copy the summaries into profiles/): the default bench line, a rocprofv3
# --kernel-trace --stats pass of the SERIAL bench (one stream: per-launch durations of overlapped streams are not a
# property of the kernel), separate --pmc passes for HBM traffic and the SQ wait counters, the per-launch step table.
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 300 python bench.py > $out/bench_line.json 2> $out/bench.err
timeout 300 python bench.py --streams 1 > $out/bench_line_serial.json 2>> $out/bench.err
timeout 300 python tools/profile_step.py > $out/step_launches.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python bench.py --streams 1 --steps 20 --warmup 3 --no-cpu-baseline > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db > $out/kernel_stats.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -o s -- python bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/pmc_s.log 2>&1
ls -R $out | head -40
