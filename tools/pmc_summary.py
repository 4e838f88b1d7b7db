"""Condense gpurun_out/prof_{stats,fetch,write,l2,dram} (tools/capture_profiles.sh) into the tracked summaries:
profiles/<round>_bench_kernel_stats.csv, profiles/<round>_pmc_summary.json, profiles/pmc_traffic.json (the
`traffic` field of bench.py's roofline object) and profiles/<round>_bench_line.json (round = $ROUND, default r2).
HBM bytes per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE tallies the 128-B
requests of 16-B-per-lane reads as 64 B, so fetched bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 as is."""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
ROUND = os.environ.get("ROUND", "r2")


def short(name):
    m = re.search(r"(spmm_vec_kernel<[^>]*>|spmm_long\w*<[^>]*>|dense_fwd_kernel<[^>]*>|dense_bwd_kernel<[^>]*>|dense_bwd_split_kernel<[^>]*>|"
                  r"reduce_partials_kernel)", name)
    return m.group(1).replace(" ", "") if m else None


def counters(directory):
    res = {}
    for path in glob.glob(os.path.join(OUT, directory, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            if k is None:
                continue
            key = (k, row["Counter_Name"], row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
        for (k, c, _), v in per.items():
            res.setdefault(k, {}).setdefault(c, []).append(v)
    return res


def main():
    merged = {}
    for d in ("prof_fetch", "prof_write", "prof_l2", "prof_dram"):
        for k, cs in counters(d).items():
            merged.setdefault(k, {}).update(cs)
    summary = {"command": "tools/capture_profiles.sh: rocprofv3 --pmc <C> --output-format csv -- python bench.py "
                          "--no-cpu-baseline --steps 3 --warmup 1 (one pass per counter group: FETCH_SIZE | WRITE_SIZE | "
                          "TCC_HIT_sum TCC_MISS_sum | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum)",
               "correction": __doc__.split("HBM bytes per")[1].strip(), "kernels": {}}
    for k, cs in sorted(merged.items()):
        ent = {c: {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()}
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            ent["hbm_bytes_per_launch_corrected"] = (ent["FETCH_SIZE"]["mean"] * 2 + ent["WRITE_SIZE"]["mean"]) * 1024
        if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
            h, m = ent["TCC_HIT_sum"]["mean"], ent["TCC_MISS_sum"]["mean"]
            ent["l2_hit_rate"] = h / (h + m) if h + m else None
        if "TCC_EA0_RDREQ_sum" in cs and "TCC_EA0_RDREQ_DRAM_sum" in cs:
            # requests the L2 sent to the memory side, and the subset routed to DRAM (the MC behind the Infinity
            # Cache -- the counter sits BEFORE the cache, so it is a routing split, not a MALL hit / miss split)
            a, d = ent["TCC_EA0_RDREQ_sum"]["mean"], ent["TCC_EA0_RDREQ_DRAM_sum"]["mean"]
            ent["ea_read_requests_routed_to_dram_fraction"] = d / a if a else None
        summary["kernels"][k] = ent
    json.dump(summary, open(os.path.join(PROF, ROUND + "_pmc_summary.json"), "w"), indent=1)
    dual = [k for k in summary["kernels"] if k.startswith("spmm_vec_kernel<16,true")]
    if dual and "hbm_bytes_per_launch_corrected" in summary["kernels"][dual[0]]:
        json.dump({"nodes": 1000000, "hidden": 64, "n_gpus": 1, "kernel": dual[0],
                   "hbm_bytes_per_launch": summary["kernels"][dual[0]]["hbm_bytes_per_launch_corrected"],
                   "source": "profiles/" + ROUND + "_pmc_summary.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes)"},
                  open(os.path.join(PROF, "pmc_traffic.json"), "w"), indent=1)
    stats = glob.glob(os.path.join(OUT, "prof_stats", "**", ROUND + "_kernel_stats.csv"), recursive=True) or \
        glob.glob(os.path.join(OUT, "prof_stats", "**", "*kernel_stats.csv"), recursive=True)      # this round's capture first
    if stats:
        shutil.copy(stats[0], os.path.join(PROF, ROUND + "_bench_kernel_stats.csv"))
    line = os.path.join(OUT, "bench_line.json")
    if os.path.exists(line) and os.path.getsize(line):
        shutil.copy(line, os.path.join(PROF, ROUND + "_bench_line.json"))
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if not isinstance(vv, dict)} for k, v in summary["kernels"].items()},
                     indent=1))


if __name__ == "__main__":
    sys.exit(main())
