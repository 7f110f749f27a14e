#!/bin/bash
# GPU box: sustained batch-mode rate of bench.py under a few library switches (each argument "KEY=V KEY2=V2" is one run)
for ENVS in "" "$@"; do
  V=$(env $ENVS python bench.py --no-extensions --no-configs --no-cpu-baseline --profile-calls 8 2>&1 >/dev/null | grep "timed region" | sed 's/.*= \([0-9.]*\) videos.*/\1/')
  echo "[${ENVS:-default}] $V videos/s"
done
