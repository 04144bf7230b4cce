#!/bin/bash
# decoder experiments: decode timing of the bench corpus per build / waves per CU
set -o pipefail
mkdir -p gpurun_out
run() {  # label, env...
    echo "=== $1" >> gpurun_out/decode_exp.txt
    shift
    env "$@" timeout -k 10 300 python -u scripts/decode_time.py 2>&1 | grep -E "QS/o1|all |1xQS|round trip|Error|error" >> gpurun_out/decode_exp.txt || return 1
}
: > gpurun_out/decode_exp.txt
run "base 5 waves/CU" A=1 &&
run "base 4 waves/CU" CRAMGPU_DEC_WAVES_PER_CU=4 &&
run "late 5 waves/CU" CRAMGPU_LIB=$PWD/cramtools_amd/libcramgpu_late.so &&
run "late 4 waves/CU" CRAMGPU_LIB=$PWD/cramtools_amd/libcramgpu_late.so CRAMGPU_DEC_WAVES_PER_CU=4
rc=$?
cat gpurun_out/decode_exp.txt
exit $rc
