#!/bin/bash
# Profiling evidence of a round, collected on the GPU box (through gpurun, from the repo root):  bash tools/collect_profiles.sh r04
# Outputs under gpurun_out/<tag>/ (the summaries are then copied into profiles/ as <tag>_*):
#   bench_line.json        the default bench line (sampler in the step, all legs, rocprofv3 --pmc traffic passes, CPU baseline)
#   bench_line_serial.json one stream
#   step_launches.txt      every launch of one C2 step (hipEvent brackets): us, GFLOP, TF/s, MB, GB/s
#   kernel_stats.txt       rocprofv3 --kernel-trace --stats of the serial C2 bench (the durations roofline.avg_launch_us must agree with)
#   pmc_sq.txt             rocprofv3 --pmc SQ_* (separate pass) of the same command
#   c3_*, c5_*             the same for the C3 probe (bench.py --probe c3) and the C5 probe (bench.py --probe c5)
#   gemm_sweep.txt, slowfast_layers.txt, attention / bias-build probes, embed_sampler_pmc.txt (HBM bytes of K1 + embedding, both sequencings)
tag=${1:-r06}
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
timeout 1200 bash tools/pmc_embed.sh > $out/embed_sampler_pmc.txt 2>&1
# attention alone on the chip (tools/ubench/*.bin are built on the build host: hipcc ... attn32_bench.hip / attn32_loop.hip) and its counters
if [ -x tools/ubench/attn32_bench.bin ]; then
  (cd tools/ubench && ./attn32_run.sh attn32_bench.bin) > $out/attn32_standalone.txt 2>&1
  (cd tools/ubench && for g in "128 3 4 392 64 20 -1" "32 6 4 392 32 20 16" "8 12 4 392 4 20 -1" "2 24 4 392 2 20 1"; do ./attn32_bench.bin $g 0 8 | tail -1; done) > $out/attn32_cold.txt 2>&1
  [ -x tools/ubench/attn32_loop.bin ] && tools/ubench/attn32_loop.bin 50 > $out/attn32_loop.txt 2>&1
  [ -x tools/ubench/attn32_loop_short.bin ] && (echo "-- the N = 385..392 form of the 13th key block (round 6, -DA32_LOOP_SHORT=1): per block of the 13, i.e. x 13 / 12.25 per USEFUL block" >> $out/attn32_loop.txt; tools/ubench/attn32_loop_short.bin 50 >> $out/attn32_loop.txt 2>&1)
  bash tools/ubench/attn32_pmc.sh attn32_bench.bin "128 3 4 392 64" $tag > /dev/null 2>&1; cp gpurun_out/attn_pmc_$tag.txt $out/attn32_pmc.txt 2>/dev/null
fi
for s in 0 1; do
  rm -rf /tmp/tr_attn; TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_attn -o t -- $C2 --steps 8 --warmup 2 --min-timed-s 0 > /dev/null 2>&1
  python tools/attn_launches.py /tmp/tr_attn >> $out/attn_launches.txt 2>&1
done
# per-launch HBM traffic of one C2 step (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes) and the family ablation of the 4-lane line
# (KVQ_SKIP is honoured by the -DKVQ_DIAG variant only: built on the build host, see tools/skip_ablation.sh)
timeout 900 python tools/step_traffic.py > $out/step_traffic.txt 2>&1
[ -f kvq-challenge-cvpr-ntire2024_amd/libkvq_hip_diag.so ] && timeout 1500 bash tools/skip_ablation.sh $out/skip_ablation.txt > /dev/null 2>&1
# the timed regime under a kernel trace: per-queue gaps, concurrency histogram, per-family launch durations in the mix
rm -rf /tmp/tr_mix; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_mix -o t -- python $OLDPWD/bench.py --probe c2mix --probe-steps 24 --streams 4 > /dev/null 2>&1)
python tools/mix_timeline.py /tmp/tr_mix 24 > $out/mix_timeline.txt 2>&1
# KSVQE forward: kernel statistics of 3 serial forwards (the one-time builders of the first forward included)
rm -rf /tmp/tr_ks; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_ks -o t -- python bench.py --probe ksvqe --probe-steps 3 > /dev/null 2>&1
db=$(find /tmp/tr_ks -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > $out/ksvqe_kernel_stats.txt
find $out -name "*.db" -size +20M -delete
find $out -name "*counter_collection.csv" -size +20M -delete
ls $out
