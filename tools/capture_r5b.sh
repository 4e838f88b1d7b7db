#!/bin/bash
# Run ON THE GPU BOX (round 5, second capture): the new kernels' tests first (signed +-1 build, piece layouts), the sharded suites,
# then the build probe and the 8-rank rehearsal in its old and new shapes.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "signed_unit or piece_layouts or unit_operator or fused_operator" > $O/r5b_pytest_new.log 2>&1; echo "rc=$?" >> $O/r5b_pytest_new.log )
tail -15 $O/r5b_pytest_new.log
( timeout 1200 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider > $O/r5b_pytest_sharded.log 2>&1; echo "rc=$?" >> $O/r5b_pytest_sharded.log )
tail -15 $O/r5b_pytest_sharded.log
timeout 300 python tools/build_probe.py --iters 10 --only fused,fused_signed,two_stage_signed,fused_real_weights > $O/r5b_build_probe.log 2>&1 && cp $O/build_probe.json $O/r5b_build_probe.json
tail -2 $O/r5b_build_probe.log | cut -c1-600
CASES=16 timeout 600 python tools/bucket_build_stress.py > $O/r5b_bucket_stress.log 2>&1; tail -3 $O/r5b_bucket_stress.log
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.92 --shapes grid:2:2 grid:0.4,0.6:0.5,0.36,0.14 grid:0.4,0.6:2 grid:2:0.5,0.36,0.14 grid:0.35,0.65:0.45,0.3,0.17,0.08 grid:0.5,0.5:0.6,0.3,0.1 --out $O/r5b_emulated_w8.json > $O/r5b_emulated_w8.log 2>&1
PYGSD_SHARD_MERGE_ON_READ=0 PYGSD_SHARD_PACKED_BACKWARD=0 timeout 300 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.92 --shapes grid:2:2 grid:0.4,0.6:0.5,0.36,0.14 --out $O/r5b_emulated_w8_noshortcuts.json > $O/r5b_emulated_w8_noshortcuts.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/r5b_emulated_w8.json", "gpurun_out/r5b_emulated_w8_noshortcuts.json"):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, e); continue
    for r in d["runs"]:
        if "error" in r: print(r); continue
        p = r["per_propagate"]
        print(f.split("_w8")[1][:14], r["phases"], r["return_chunks"], "step %.3f" % r["step_ms_median"], "prop %.3f pack %.3f in %.3f prod %.3f out %.3f merge %.3f" % (p["total_ms"], p["pack_ms"], p["wait_in_ms"], p["product_ms"], p["wait_out_ms"], p["merge_ms"]), r.get("merge_on_read"), r.get("packed_backward"))
PY
