#!/bin/bash
# GPU box: the driver-style timed region of bench.py (20 steps x 1536 videos, sustained) for a few (streams, videos per launch set, videos per call)
for CFG in "3 8 48" "4 4 48" "4 4 96" "3 8 96" "4 8 96" "2 8 96" "4 4 192" "3 16 96"; do
  set -- $CFG
  V=$(STTM_BATCH_STREAMS=$1 STTM_BATCH_SUB=$2 python bench.py --batch $3 --no-extensions --no-configs --no-cpu-baseline --profile-calls 8 2>&1 >/dev/null | grep "timed region" | sed 's/.*= \([0-9.]*\) videos.*/\1/')
  echo "streams $1 x $2 per set, $3 per call: $V videos/s"
done
