"""Condense the operator-build captures (ROUND=r3: tools/capture_r3.sh -> gpurun_out/r3_prof_build, r3_pmc_build_fetch / _write;
ROUND=r4: tools/capture_r4_final.sh -> gpurun_out/r4_prof_build, r4_pmc_build_FETCH_SIZE / _WRITE_SIZE: rocprofv3 --kernel-trace
--stats and separate --pmc FETCH_SIZE / WRITE_SIZE passes of the same command) into profiles/<round>_build_kernel_stats.csv: per kernel of ONE fused north-star build (unweighted leg), average duration, launches
per build, FETCH_SIZE / WRITE_SIZE per launch and the bytes the kernel has to move by its algorithm.
HBM bytes per MI355X_MICROARCH.md: the counters are in KiB; FETCH_SIZE tallies the 128-byte requests of wide (8 / 16 bytes
per lane) reads as 64 bytes on gfx950, so fetched bytes = FETCH_SIZE x 1024 x 2 (marked `x2`; narrower accesses are
uncalibrated -- the raw figure is kept next to it); WRITE_SIZE x 1024 as is."""
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
ROUND = os.environ.get("ROUND", "r3")
NAMES = ["edge_keys", "onesweep_histograms", "onesweep_iteration", "key_row_starts", "row_merge_wave", "row_merge_block",
         "lookback_scan", "init_lookback", "unit_row_tables", "unit_merge_rows", "unit_write_rows", "row_tables", "values_entries",
         "diagonal_of_empty_rows"]


def short(name):
    for nm in NAMES:
        if nm in name:
            return nm
    if "radix_sort" in name or "rocprim" in name:
        m = re.search(r"(radix_sort_\w+|block_sort\w*|scan\w*)", name)
        return "rocprim:" + (m.group(1) if m else name[:40])
    return None


def counters(directory, counter):
    res = {}
    for path in glob.glob(os.path.join(OUT, directory, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            if k:
                per[(k, row["Dispatch_Id"])] = per.get((k, row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
        for (k, _), v in per.items():
            res.setdefault(k, []).append(v)
    return {k: sum(v) / len(v) for k, v in res.items()}


def main():
    probe = json.load(open(os.path.join(OUT, ROUND + "_build_probe.json")))
    n, e, nnz = probe["nodes"], probe["edges"], probe["operator_nnz"]
    m = 2 * e
    # what each kernel must move (unweighted leg): bytes in + out by the algorithm
    algorithmic = {
        "edge_keys": 16 * e + 8 * m,                       # int64 COO in, u64 keys out
        "onesweep_histograms": 8 * m,
        "onesweep_iteration": 16 * m,                      # per pass: keys in + keys out
        "key_row_starts": 8 * m + 4 * n,
        "row_merge_wave": 8 * m + 16 * m + 12 * n,         # keys in, 16-byte records out, per-row count / degree
        "row_tables": 16 * n,
        "values_entries": 16 * m + 20 * nnz + 12 * n,      # records in, col + 4 value arrays out (+ row tables, gathers extra)
        # round 4, unweighted: merged rows as 8-byte records (one per distinct neighbour), read back after the scan
        "unit_row_tables": 12 * n,
        "unit_merge_rows": 8 * m + 8 * (nnz - n) + 12 * n,
        "unit_write_rows": 8 * (nnz - n) + 20 * nnz + 16 * n,
    }
    stats = {}
    for path in glob.glob(os.path.join(OUT, ROUND + "_prof_build", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Name"])
            if k:
                rec = stats.setdefault(k, {"calls": 0, "total_ns": 0.0})
                rec["calls"] += int(row["Calls"])
                rec["total_ns"] += float(row["TotalDurationNs"])
    names = ("r3_pmc_build_fetch", "r3_pmc_build_write") if ROUND == "r3" else (ROUND + "_pmc_build_FETCH_SIZE", ROUND + "_pmc_build_WRITE_SIZE")
    fetch, write = counters(names[0], "FETCH_SIZE"), counters(names[1], "WRITE_SIZE")
    os.makedirs(PROF, exist_ok=True)
    with open(os.path.join(PROF, ROUND + "_build_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls_in_trace", "avg_us", "FETCH_SIZE_KiB_per_launch", "fetched_MB_x2", "WRITE_SIZE_KiB_per_launch",
                    "written_MB", "algorithmic_MB", "algorithmic_GBps_at_avg"])
        for k, rec in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"]):
            avg = rec["total_ns"] / rec["calls"]
            f, wr, alg = fetch.get(k), write.get(k), algorithmic.get(k)
            w.writerow([k, rec["calls"], round(avg / 1e3, 1), None if f is None else round(f, 1),
                        None if f is None else round(f * 1024 * 2 / 1e6, 1), None if wr is None else round(wr, 1),
                        None if wr is None else round(wr * 1024 / 1e6, 1), None if alg is None else round(alg / 1e6, 1),
                        None if alg is None else round(alg / avg, 1)])
    json.dump(probe, open(os.path.join(PROF, ROUND + "_build_probe.json"), "w"), indent=1)
    print(open(os.path.join(PROF, ROUND + "_build_kernel_stats.csv")).read())


if __name__ == "__main__":
    main()
