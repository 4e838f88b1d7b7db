#!/bin/bash
# GPU box: SQ / LDS counters of the unweighted operator build's kernels (separate --pmc passes, no trace domains)
TAG=${TAG:-r4}
cd /root/repo; export TMPDIR=/tmp
i=0
for set in ${PMC_SETS:-"FETCH_SIZE" "WRITE_SIZE"}; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_build_${TAG}_$i
  timeout 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_build_${TAG}_$i -o b -- python tools/build_probe.py --only fused --iters 2 > gpurun_out/pmc_build_${TAG}_$i.log 2>&1
done
python - <<P
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_build_${TAG}_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pygsd" not in k: continue
        import re
        m = re.search(r"namespace\)::(\w+)", k)
        short = m.group(1) if m else k[:40]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v)/len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("gpurun_out/${TAG}_build_pmc.json", "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:26s} {v:16.0f}")
P
