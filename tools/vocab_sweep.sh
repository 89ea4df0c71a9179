#!/bin/bash
# zeroshot_classifier over 20,480 names x 80 templates: names per batch x length groups (bench.py --config c5's numbered names)
export VB_NAMES=numbered
for cfg in "256 4 2048" "1024 4 2048" "1024 8 2048" "1024 16 2048" "2048 16 2048"; do
  set -- $cfg
  VB_GROUPS=$2 VB_MIN_GROUP=$3 timeout -k 10 200 python tools/vocab_bench.py 20480 $1 2>/dev/null | tail -1 | cut -c1-120 | sed "s/^/groups $2: /"
done
