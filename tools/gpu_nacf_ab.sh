#!/bin/bash
# NAcF alone at M = 2048 / 6464, RAF and SoundSpaces heads, row padding 128 vs 256 (tools/nacf_bench.py)
for al in 128 256; do
  echo "=== NERAF_NACF_MALIGN=$al"
  NERAF_NACF_MALIGN=$al python tools/nacf_bench.py 2048 6464 2>&1 | grep -v amdgpu.ids
  NERAF_NACF_MALIGN=$al python tools/nacf_bench.py 6464 --ss 2>&1 | grep -v amdgpu.ids
done
