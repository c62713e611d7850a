set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bench; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 400 python3 -u bench.py --steps 100 --warmup 10 > $O/bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
tail -c 3000 $O/bench_line.json
QTTS_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -u bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2rank_rehearsal.json 2> $O/bench2.err || { tail -20 $O/bench2.err; exit 1; }
tail -c 600 $O/bench_2rank_rehearsal.json
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/trace.err || { tail -20 $O/trace.err; exit 1; }
python3 $R/tools/profile_summary.py $O/trace/* $O/summary.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline" | head -30
