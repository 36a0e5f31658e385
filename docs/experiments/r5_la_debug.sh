#!/bin/bash
out=gpurun_out/r05_lookahead_debug.txt
: > $out
for v in ${LA_VARIANTS:-"" "sync" "eager-bl" "eager-el" "eager-p" "streams" "plain-after"}; do
  echo "=== variant: $v" >> $out
  timeout -k 10 300 python tools/debug_lookahead_single.py $v 2>&1 | grep -v amdgpu.ids | tail -${LA_TAIL:-25} >> $out
done
cat $out
