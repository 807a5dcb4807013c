#!/bin/bash
# Round-6 call 34: which change made the bench's self-check flaky: 4 short bench runs (self-check on) per library
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_aq_flaky2.txt; : > $O
cp ipercore_amd/liblwg_hip.so /tmp/liblwg_tree.so
for v in tree commit_xcd commit_lds commit_aj; do
  if [ $v = tree ]; then cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so; else cp tools/lab/liblwg_$v.so ipercore_amd/liblwg_hip.so; fi
  for i in 1 2 3 4; do
    timeout 300 python bench.py --steps 2 --warmup 1 --no-extras --cpu-frames 0 > /tmp/b.out 2> /tmp/b.err
    if grep -q '"self_check": "bitwise"' /tmp/b.out; then echo "$v run $i: bitwise" >> $O; else echo "$v run $i: FAILED $(grep -o 'self-check failed.*' /tmp/b.err | head -1 | cut -c1-200)" >> $O; fi
  done
done
cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so
cat $O
