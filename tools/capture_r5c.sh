#!/bin/bash
# Run ON THE GPU BOX (round 5, third capture): gram 32x32 tests, the sharded suites with the new defaults, the rehearsal with the
# host-issue time and a hipGraph replay per shape, C3a / C5a with the weight-gradient product in its 32x32 form and in round 4's.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "tall_gram or tall_product or tall_linear" > $O/r5c_pytest_gram.log 2>&1; echo "rc=$?" >> $O/r5c_pytest_gram.log )
tail -12 $O/r5c_pytest_gram.log
( timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider > $O/r5c_pytest_sharded.log 2>&1; echo "rc=$?" >> $O/r5c_pytest_sharded.log )
tail -12 $O/r5c_pytest_sharded.log
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.92 --shapes grid:2:2 grid:0.4,0.6:2 grid:1:1 grid:0.4,0.6:0.3,0.7 --out $O/r5c_emulated_w8.json > $O/r5c_emulated_w8.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c_emulated_w8.json"))
for r in d["runs"]:
    if "error" in r: print(r); continue
    p = r["per_propagate"]
    print(r["phases"], r["return_chunks"], "step %.3f host %.3f graph %s" % (r["step_ms_median"], r["host_issue_ms_per_step"], r["step_ms_hipgraph_replay"]), (r.get("hipgraph_error") or "")[:150],
          "prop %.3f pack %.3f in %.3f prod %.3f out %.3f merge %.3f" % (p["total_ms"], p["pack_ms"], p["wait_in_ms"], p["product_ms"], p["wait_out_ms"], p["merge_ms"]))
PY
for v in 1 0; do
  PYGSD_GRAM_32X32=$v PYGSD_CONFIGS=C3a,C5a timeout 400 python tools/bench_configs.py > $O/r5c_configs_gram$v.log 2>&1; cp $O/configs_partial.json $O/r5c_configs_gram$v.json
  echo "gram32=$v"; grep -E "^C3a|^C5a|dense_bwd" $O/r5c_configs_gram$v.log | cut -c1-300 | head -8
done
