#!/bin/bash
# two INDEPENDENT single-process trajectory runs sharing the GPU at the same time (no torch.distributed): does sharing alone disturb them?
for i in 1 2 3 4 5 6; do
  timeout 300 python tools/half_batch_check.py 2>/dev/null | tail -1 > /tmp/sg_a.txt &
  timeout 300 python tools/half_batch_check.py 2>/dev/null | tail -1 > /tmp/sg_b.txt &
  wait
  echo "pair $i: $(cat /tmp/sg_a.txt) | $(cat /tmp/sg_b.txt)"
done
