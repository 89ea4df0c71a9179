#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gputest.txt 2>&1; rc=$?
tail -n 5 $O/gputest.txt
exit $rc
