#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() {
    echo "=== $1" >> gpurun_out/decode_exp.txt
    shift
    env "$@" timeout -k 10 300 python -u scripts/decode_time.py 2>&1 | grep -E "QS/o1|TS/o1|all |1xQS|round trip|Error|error" >> gpurun_out/decode_exp.txt || return 1
}
: > gpurun_out/decode_exp.txt
run "6 waves/CU" CRAMGPU_DEC_WAVES_PER_CU=6 &&
run "5 waves/CU" CRAMGPU_DEC_WAVES_PER_CU=5
rc=$?
cat gpurun_out/decode_exp.txt
exit $rc
