"""Measurement helper (GPU box host): bench.py's CPU baseline leg at several thread counts -- ATen's
index_select / scatter_add_ stop scaling long before a 256-thread host is full, so bench.py pins 32."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for th in (8, 16, 32, 64, 128):
    r = bench.cpu_baseline(64, steps=1, threads=th)
    print(th, json.dumps({"edges_per_s": r["value"], "sample": r["sample"][-40:]}), flush=True)
