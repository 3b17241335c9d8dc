#!/bin/bash
# same-box A/B of the sampler-folded depth range (round 6) against the composite's own seed + reduction launches: eval frames and the training step
for round in 1 2 3; do for f in 1 0; do
  NERAF_MINMAX_FOLD=$f python bench.py --mode eval --steps 20 --warmup 3 --plain --rirs 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eval  fold=$f', 'ms_per_frame(step, 0 RIRs)', round(d['ms_per_step'],4))"
done; done
for round in 1 2; do for f in 1 0; do
  NERAF_MINMAX_FOLD=$f python bench.py --steps 30 --warmup 5 --plain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train fold=$f', 'ms_per_step', round(d['ms_per_step'],4))"
done; done
