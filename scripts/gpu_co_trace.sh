#!/bin/bash
# kernel trace of the coalescing queue under 64 outstanding tickets: which hardware queue each launch used, what overlapped
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export CO_THREADS=${CO_THREADS-} CO_TICKETS=${CO_TICKETS-64}
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/co_ktrace -- python3 -u $GRAFT_REPO_ROOT/scripts/coalesce_time.py 128 4 > $GRAFT_REPO_ROOT/gpurun_out/co_ktrace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/co_ktrace.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/co_ktrace/*/*kernel_trace.csv | head -1)
head -1 $f > gpurun_out/co_ktrace_enc.csv
grep -E "rans_encode|gather" $f | tail -80 >> gpurun_out/co_ktrace_enc.csv
rm -rf gpurun_out/co_ktrace
tail -3 gpurun_out/co_ktrace.log
