#!/bin/bash
# Runs on the GPU box: the measurements kept under profiles/ (bench line, kernel trace summary, PMC passes,
# API-level examples, prefill table).  Output goes to gpurun_out/refresh/ (the files for profiles/ under
# gpurun_out/refresh/profiles/); every step has its own time limit.
set -e -o pipefail
R=$GRAFT_REPO_ROOT
RND=r03
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
step() { echo "== $1" >&2; }

step pmc-fetch
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/microbench.py dominant >> $O/prof.log 2>&1
step pmc-write
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/microbench.py dominant >> $O/prof.log 2>&1
step pmc-predictor
QTTS_DOMINANT=predictor timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_pred -- python3 $R/tools/microbench.py dominant >> $O/prof.log 2>&1
QTTS_DOMINANT=predictor timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_pred -- python3 $R/tools/microbench.py dominant >> $O/prof.log 2>&1
python3 $R/tools/collect_profiles.py --traffic-only
step bench-trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/trace.err
# the bench line quotes the profiler's figure for its dominant kernel from profiles/: put this trace's summary there first
rm -f $R/profiles/${RND}_bench_summary_*.txt
TD=($O/trace/*)
python3 $R/tools/profile_summary.py ${TD[0]} $R/profiles/${RND}_bench_summary_${1:-v1}.txt "same box, same build" > /dev/null
step bench
timeout -k 10 400 python3 -u $R/bench.py --steps 100 --warmup 10 > $O/bench_line.json 2> $O/bench.err
step pmc-mfma
QTTS_ROWS=200 timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/microbench.py rows >> $O/prof.log 2>&1
step prefill
QTTS_FRAME_PROMPT=16,48,128,200,400,900 timeout -k 10 300 python3 -u $R/tools/microbench.py prefill > $O/prefill.txt 2>> $O/err.log
QTTS_FRAME_PROMPT=200 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prefill_trace -- python3 $R/tools/microbench.py prefill >> $O/prof.log 2>&1
step arrivals-admission
cd $R
for m in serial beside; do
  QTTS_ADMISSION=$m timeout -k 10 300 python3 -u bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_admission_$m.json 2>> $O/err.log
done
step two-ranks
QTTS_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -u bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2rank_rehearsal.json 2>> $O/err.log
step configs
# BASELINE.json's other configurations as bench lines of their own: 0.6B shapes with one stream; fp8 weights at 8 streams
timeout -k 10 300 python3 -u bench.py --size 0.6B --streams 1 --steps 100 --no-cpu-baseline > $O/config_0.6B_1stream.json 2>> $O/err.log
timeout -k 10 300 python3 -u bench.py --weights fp8 --steps 100 --no-cpu-baseline > $O/config_fp8_8streams.json 2>> $O/err.log
step examples
cd $R
timeout -k 10 300 python3 -u examples/quick_benchmark.py > $O/api_quick_benchmark.json 2> $O/qb.err
timeout -k 10 300 python3 -u examples/batch_async_example.py --streams 8 > $O/api_batch_async_fused.json 2> $O/ba.err
timeout -k 10 300 python3 -u examples/batch_async_example.py --streams 8 --engine workers > $O/api_batch_async_workers.json 2> $O/bw.err
timeout -k 10 300 python3 -u examples/quick_benchmark.py --sample > $O/api_quick_benchmark_sampling.json 2> $O/qbs.err
timeout -k 10 300 python3 -u examples/batch_async_example.py --streams 8 --sample > $O/api_batch_async_fused_sampling.json 2> $O/bas.err
step sweep
for S in 1 16 24 32 64 128 256; do
  timeout -k 10 300 python3 -u bench.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/sweep_$S.json 2>> $O/err.log
done
step trace-32-streams
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace32 -- python3 $R/bench.py --streams 32 --steps 30 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_32_profiled.json 2>> $O/trace.err
cd $R
step soak
QTTS_SOAK_STEPS=1500 QTTS_SOAK_REPLAY_STEPS=500 timeout -k 10 500 python3 -u tools/soak.py > $O/soak.json 2>> $O/err.log
QTTS_SOAK_STREAMS=32 QTTS_SOAK_STEPS=1500 QTTS_SOAK_REPLAY_STEPS=500 timeout -k 10 500 python3 -u tools/soak.py > $O/soak_32rows.json 2>> $O/err.log
step rows
QTTS_ROWS=32,64,128,200,400 timeout -k 10 300 python3 -u tools/microbench.py rows > $O/rows.txt 2>> $O/err.log
step collect
# the raw traces are hundreds of MB: reduce them here, keep only the summaries
QTTS_PROFILES_OUT=$O/profiles python3 tools/collect_profiles.py ${1:-v1} > $O/collect.txt 2> $O/collect.err
rm -rf $O/trace $O/trace32 $O/pmc_fetch $O/pmc_write $O/pmc_fetch_pred $O/pmc_write_pred $O/pmc_mfma $O/prefill_trace
echo done >&2
