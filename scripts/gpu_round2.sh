#!/bin/bash
# round 2 evidence: bench line, kernel trace of the same command, HBM traffic passes, SQ counter pass
set -o pipefail
bash scripts/gpu_prof.sh || exit 1
bash scripts/gpu_pmc.sh || exit 1
bash scripts/gpu_sq.sh after > gpurun_out/sq_after.txt 2>&1 || { tail -20 gpurun_out/sq_after.txt; exit 1; }
cat gpurun_out/sq_after.txt
