#!/bin/bash
# Run each GPU test in its own process (a device fault in one must not hide the others' reports).
mkdir -p gpurun_out
: > gpurun_out/each.log
for t in $(python -m pytest tests -m gpu --collect-only -q -p no:cacheprovider 2>/dev/null | grep "::"); do
  if [ -n "$1" ] && ! echo "$t" | grep -q -E "$1"; then continue; fi
  echo "=== $t" >> gpurun_out/each.log
  timeout 300 python -m pytest "$t" -x -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v "^  File \"/usr\|^$" | tail -40 >> gpurun_out/each.log
done
grep -E "^=== |passed|failed|Abort|fault|Error|assert|mismatch|\[" gpurun_out/each.log | tail -150
