#!/bin/bash
# GPU session 6 of round 3: arrangement probe (same chunks, other order).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A=tools/experiments/bin/arrange_probe
{
  echo "=== C3 size, 32 MiB chunks"; timeout 200 $A 3616 32 4 0
  echo "=== C3 size, 32 MiB chunks, 7 KB LDS pad"; timeout 200 $A 3616 32 2 7168
  echo "=== C4 size, 32 MiB chunks"; timeout 200 $A 4288 32 3 0
  echo "=== C3 size, 256 MiB chunks"; timeout 200 $A 3616 256 2 0
  echo "=== C3 size, 2 MiB chunks"; timeout 300 $A 3616 2 2 0
} > $O/r03_arrange.txt 2>&1
tail -n 30 $O/r03_arrange.txt
