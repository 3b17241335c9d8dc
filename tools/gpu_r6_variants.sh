#!/bin/bash
# Round 6: the joint step with the non-default encoder configurations (bench.py --plain --grid / --n-features) + the pipeline test.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s -k "other_encoder" 2>&1 | grep -v "^$" | tail -6 | cut -c1-250
for cfg in "--grid 128 --n-features 1024" "--grid 128 --n-features 2048" "--grid 256 --n-features 1024" "--grid 256 --n-features 2048"; do
  echo "== $cfg"
  timeout 600 python bench.py --plain --steps 30 --warmup 3 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d['config'].get('encoder_grid'), d['config'].get('encoder_features'))"
done
