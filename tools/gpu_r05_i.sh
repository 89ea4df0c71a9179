#!/bin/bash
set -u
R=$PWD
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
bash tools/gpu_prof_any.sh lloyd_merged tools/lloyd_multi_prof.py 3
SCD_ESTEP_MERGED=0 bash tools/gpu_prof_any.sh lloyd_unmerged tools/lloyd_multi_prof.py 3
