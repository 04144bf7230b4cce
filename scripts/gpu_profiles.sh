#!/bin/bash
bash scripts/gpu_prof.sh || exit 1
bash scripts/gpu_pmc.sh || exit 1
