#!/bin/bash
# Counters of the attention kernels: matrix-core busy cycles of the prompt attention (900- and 200-token prompts), HBM
# fetch of the decode attention in eagerly launched frames at 8 streams (contexts of about 300 tokens); durations from a plain kernel trace of the same commands.
# Separate --pmc passes with --kernel-trace only.  Output: gpurun_out/attn_pmc/r02_pmc_attention.json
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
QTTS_FRAME_PROMPT=200,900 timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $O/pre_trace -- python3 $R/tools/microbench.py prefill > $O/log.txt 2>&1
QTTS_FRAME_PROMPT=200,900 timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pre_mfma -- python3 $R/tools/microbench.py prefill >> $O/log.txt 2>&1
QTTS_FRAME_PROMPT=200,900 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pre_fetch -- python3 $R/tools/microbench.py prefill >> $O/log.txt 2>&1
# (frames launched eagerly: counters are collected per launch, and the profiler's counter pass was seen to crash in the host
# process on graph replays)
QTTS_FRAME_PROMPT=300 timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/dec_trace -- python3 $R/tools/microbench.py decode_eager >> $O/log.txt 2>&1
QTTS_FRAME_PROMPT=300 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/dec_fetch -- python3 $R/tools/microbench.py decode_eager >> $O/log.txt 2>&1
cd $R
python3 tools/attn_pmc_summary.py $O > $O/summary.txt 2>&1 || { tail -20 $O/summary.txt; exit 1; }
cat $O/summary.txt
rm -rf $O/pre_trace $O/pre_mfma $O/pre_fetch $O/dec_trace $O/dec_fetch
