#!/bin/bash
# upper bound of sorting the training field query's samples (tools/gather_sort_bound.py) + FETCH_SIZE per order
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/sort
cd $R
timeout 600 python3 tools/gather_sort_bound.py > gpurun_out/sort/bound.json 2> gpurun_out/sort/bound.err; echo "bound rc=$?"
cat gpurun_out/sort/bound.json
cd /tmp
for o in ray morton random; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/sort/pmc_$o -- python3 $R/tools/gather_sort_bound.py --order $o --pmc > $R/gpurun_out/sort/pmc_$o.log 2>&1
  echo "pmc $o rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, json
out = {}
for o in ("ray", "morton", "random"):
    fs = glob.glob(f"gpurun_out/sort/pmc_{o}/**/*_counter_collection.csv", recursive=True)
    v = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "field_query_kernel" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    # the N-sample single-ray launches are the last 21 of each process (the sampler chain's structured call precedes them)
    v = v[-21:]
    out[o] = {"launches": len(v), "fetch_MB_per_launch": round(sum(v) / max(len(v), 1) * 1024 * 2 / 1e6, 1)}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/sort/fetch.json", "w"), indent=1)
PY
rm -rf gpurun_out/sort/pmc_ray gpurun_out/sort/pmc_morton gpurun_out/sort/pmc_random
