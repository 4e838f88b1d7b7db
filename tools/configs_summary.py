"""Condense a per-configuration capture (tools/capture_r4b.sh: one rocprofv3 --kernel-trace --stats run and one --pmc FETCH_SIZE /
WRITE_SIZE pass per configuration step, each step alone in its process) into profiles/<tag>_configs.json:

  per configuration, per HIP kernel of this library:  calls per step, average duration (kernel_stats.csv, the un-instrumented
  run), ALGORITHMIC bytes per launch (SURVEY.md 8(d) byte model: every operand read once, every result written once, no reuse
  credit) or flops, the rate they give and its fraction of the 8 TB/s HBM peak / 157 TF fp32 (2.5 PF bf16) MFMA peak, and the
  MEASURED bytes per launch from the counters (FETCH_SIZE x 1024 x 2 [gfx950: 128-byte requests of 16-byte-per-lane reads are
  tallied as 64 bytes, MI355X_MICROARCH.md] + WRITE_SIZE x 1024; memory-side L2 traffic: Infinity-Cache hits included).

and copies the kernel_stats.csv of every step to profiles/<tag>_kernel_stats_<step>.csv.   usage: configs_summary.py [tag]"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r4b"
STEPS = {"C3a": ("C3_sgcnconv_first", 13), "C3b": ("C3_simpa_hop2", 13), "C5a": ("C5_digcn_inception_block_1gpu", 14),
         "C5b": ("C5_digcn_inception_block_1gpu", 14)}          # (key in configs json, timed + warm-up steps run by the tool)
NAME = re.compile(r"(spmm_\w+<[^>]*>|tall_linear_\w+<[^>]*>|tall_gram32_\w+<[^>]*>|tall_gram_\w+|gram_finish_kernel|gemm_\w+|column_sums\w*(?:<[^>]*>)?|"
                  r"dots_kernel|weighted_sum_kernel|Cijk_\w{0,40})")


def short(name):
    m = NAME.search(name)
    return m.group(1).replace(" ", "") if m else None


def spmm_bytes(nnz, n, f, s=4, val=True, z=False):
    return nnz * (4 + (4 if val else 0) + f * s) + n * f * s * (2 if z else 1) + 4 * (n + 1)


def model(step, cfg, present=()):
    """{kernel short name: (algorithmic bytes per launch averaged over the step's launches of that kernel, flops or None, what)}"""
    if step == "C3a":
        n, pos, neg = cfg["nodes"], cfg["pos_entries"], cfg["neg_entries"]
        fwd_p, fwd_n = spmm_bytes(pos, n, 32, val=False, z=True), spmm_bytes(neg, n, 32, val=False, z=True)
        bwd_p, bwd_n = spmm_bytes(pos, n, 32), spmm_bytes(neg, n, 32)
        both = "spmm_vec_kernel<8,false,false>" not in present       # round 4's crossover packs the 14.6-entry rows too
        spmm = ({"spmm_packed_kernel<8,false>": ((fwd_p + bwd_p + fwd_n + bwd_n) / 4, None, "positive (5.4 entries per row) and negative "
                 "(14.6) part, width 32: value-less mean forward (+ own block as Z), weighted backward")} if both else
                {"spmm_packed_kernel<8,false>": ((fwd_p + bwd_p) / 2, None, "positive part (5.4 entries per row), width 32: value-less mean forward (+ own block as Z), weighted backward"),
                 "spmm_vec_kernel<8,false,false>": ((fwd_n + bwd_n) / 2, None, "negative part (14.6 entries per row), width 32, forward + backward")})
        return {**spmm,
                "tall_linear_f32_kernel<4,8>": (n * (64 + 128) * 4, 2 * n * 64 * 128, "x [own_b | own_u | agg_b | agg_u]"),
                "tall_linear_f32_kernel<8,4>": (n * (128 + 64) * 4, 2 * n * 128 * 64, "dx = [g | g_a] W^T"),
                # round 5's split form (k-blocks of 32): the same products, flops counted as fp32 multiply-adds of the operands
                "tall_linear_f32_split_kernel<2,8>": (n * (64 + 128) * 4, 2 * n * 64 * 128, "x [own_b | own_u | agg_b | agg_u] (split form)"),
                "tall_linear_f32_split_kernel<4,4>": (n * (128 + 64) * 4, 2 * n * 128 * 64, "dx = [g | g_a] W^T (split form)"),
                **{k: (n * (64 + 128) * 4, 2 * n * 64 * 128, "dW = x^T [g | g_a]") for k in present if k and k.startswith("tall_gram")},
                "column_sums_kernel<false>": (n * 64 * 4, None, "bias gradient")}
    if step == "C3b":
        n, h = cfg["nodes"], cfg["hidden"]
        res = {}
        if "pos_entries" in cfg:
            # operators of conv_norm_rw: the part's entries + one (re-)added loop per node; 4 of the 6 products of a
            # direction run on A_p, 2 on A_n; the kernel variant follows entries per row (launch_spmm)
            a_p, a_n = cfg["pos_entries"] + n, cfg["neg_entries"] + n
            for nnz, calls in ((a_p, 8), (a_n, 4)):
                per_row = nnz / n
                name = "spmm_packed_kernel<16,false>" if per_row < 10 else "spmm_vec_kernel<16,false,false>"
                b, c0, _ = res.get(name, (0.0, 0, ""))
                res[name] = (b + spmm_bytes(nnz, n, h) * calls, c0 + calls, "")
            res = {k: (b / c, None, f"A_p / A_n products at width {h} ({c} per step; Z operand not counted)") for k, (b, c, _) in res.items()}
        # (rounds 3-4 priced one of the two launches at 2 terms in: hop 2 has hop + 1 = 3 terms for feat_p AND (1 + hop) hop / 2 = 3
        # for feat_n -- SIMPA.py:60-61 -- so both read 3 matrices and write 1; the "1.14 x algorithmic" of r4j was this model)
        res.update({"weighted_sum_kernel": (4 * n * h * 4, None, "feat = sum_h w[h] cur_h (3 terms in, 1 out; either half)"),
                "dots_kernel": (4 * n * h * 4, None, "hop-weight gradients <g, cur_h>: g and 3 terms read once")})
        return res
    n, nnz = cfg["nodes"], cfg["nnz"]
    s = 4 if step == "C5a" else 2
    t = "f32" if step == "C5a" else "bf16"
    lin = {"C5a": ("tall_linear_f32_kernel<4,12>", "tall_linear_f32_kernel<12,4>"),
           "C5b": ("tall_linear_bf16_kernel<2,12>", "tall_linear_bf16_kernel<6,4>")}[step]
    return {("spmm_vec_kernel<16,false,false>" if step == "C5a" else "spmm_vec_bf16_kernel<8>"):
            (spmm_bytes(nnz, n, 64, s), None, "S_k^T P_k forward and S_k dx_k backward, 26 entries per row"),
            lin[0]: (n * (64 + 192) * s, 2 * n * 64 * 192, "x [W_ln^T | W_1 | W_2]"),
            lin[1]: (n * (192 + 64) * s, 2 * n * 192 * 64, "dx = [dx0 | dP_1 | dP_2] W^T"),
            **({"tall_linear_f32_split_kernel<2,12>": (n * (64 + 192) * 4, 2 * n * 64 * 192, "x [W_ln^T | W_1 | W_2] (split form)"),
                "tall_linear_f32_split_kernel<6,4>": (n * (192 + 64) * 4, 2 * n * 192 * 64, "dx = [dx0 | dP_1 | dP_2] W^T (split form)")}
               if step == "C5a" else {}),
            **{k: (n * (64 + 192) * s, 2 * n * 64 * 192, "x^T [dx0 | dP_1 | dP_2]") for k in present if k and k.startswith("tall_gram")},
            f"column_sums_kernel<{'false' if step == 'C5a' else 'true'}>": (n * 64 * s, None, "bias gradients")}


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
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}


def main():
    out = {"capture": f"tools/capture_{TAG}.sh on one MI355X; per step: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and "
                      "--pmc WRITE_SIZE (separate passes, counters only) over `python tools/bench_configs.py` with "
                      "PYGSD_CONFIGS=<step>", "peaks": {"hbm_GBps": 8000, "mfma_f32_TF": 157, "mfma_bf16_TF": 2500},
           "byte_model": __doc__.split("ALGORITHMIC bytes per launch")[1].split("and copies")[0].strip(), "steps": {}}
    for step, (key, runs) in STEPS.items():
        cfg_path = os.path.join(OUT, f"{TAG}_configs_{step}.json")
        stats = glob.glob(os.path.join(OUT, f"{TAG}_prof_{step}", "**", "*kernel_stats.csv"), recursive=True)
        if not (os.path.exists(cfg_path) and stats):
            continue
        shutil.copy(stats[0], os.path.join(PROF, f"{TAG}_kernel_stats_{step}.csv"))
        cfg = json.load(open(cfg_path))[key]
        if step.startswith("C5"):
            sub = cfg["float32" if step == "C5a" else "bfloat16"]
            cfg = dict(cfg, nnz=sub["nnz_per_operator"], **{k: v for k, v in sub.items() if k.startswith("ms_")})
        pmc = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum", "SQ_INSTS_VALU_SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES_SQ_WAVES"):
            for k, v in counters(f"{TAG}_pmc_{step}_{c}").items():
                pmc.setdefault(k, {}).update(v)
        names = {short(r["Name"]) for r in csv.DictReader(open(stats[0]))}
        # steps the trace covers: the timed + warm-up steps AND (since the replayed variant is traced too) the hipGraph warm-ups,
        # capture and replays of tools/bench_configs.py -- read off the SpMM launches, whose number per step is known
        spmm_calls = sum(int(r["Calls"]) for r in csv.DictReader(open(stats[0])) if (short(r["Name"]) or "").startswith("spmm_"))
        per_step = {"C3a": 4, "C3b": 12, "C5a": 4, "C5b": 4}[step]
        if spmm_calls and spmm_calls % per_step == 0:
            runs = spmm_calls // per_step
        alg = model(step, cfg, names)
        kernels, library = {}, {}
        for row in csv.DictReader(open(stats[0])):
            k = short(row["Name"])
            if k is None:
                continue
            rec = {"calls_per_step": int(row["Calls"]) / runs, "avg_us": float(row["AverageNs"]) / 1e3}
            if k.startswith("Cijk_"):
                library[k] = rec
                continue
            c = pmc.get(k, {})
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                rec["measured_bytes_per_launch"] = (c["FETCH_SIZE"] * 2 + c["WRITE_SIZE"]) * 1024
            if "TCC_HIT_sum" in c:
                rec["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
                rec["valu_instructions_per_wavefront"] = c["SQ_INSTS_VALU"] / max(1.0, c["SQ_WAVES"])
            if k in alg:
                b, fl, what = alg[k]
                rec.update(what=what, algorithmic_bytes_per_launch=b, algorithmic_GBps=b / rec["avg_us"] / 1e3,
                           fraction_of_8TBps=b / rec["avg_us"] / 1e3 / 8000)
                if "measured_bytes_per_launch" in rec:
                    rec["measured_over_algorithmic"] = rec["measured_bytes_per_launch"] / b
                if fl:
                    peak = 2500 if "bf16" in k else 157
                    rec.update(flops_per_launch=fl, TFLOPs=fl / rec["avg_us"] / 1e6, fraction_of_mfma_peak=fl / rec["avg_us"] / 1e6 / peak)
            kernels[k] = rec
        hip_us = sum(r["calls_per_step"] * r["avg_us"] for r in kernels.values())
        out["steps"][step] = {"config": {k: v for k, v in cfg.items() if not isinstance(v, dict) and k != "note"},
                              "kernels": kernels, "library_gemm_kernels": library,
                              "hip_kernel_us_per_step": hip_us}
    path = os.path.join(PROF, f"{TAG}_configs.json")
    json.dump(out, open(path, "w"), indent=1)
    for step, rec in out["steps"].items():
        print(step, {k: v for k, v in rec["config"].items() if k.startswith("ms_")}, "HIP kernels %.0f us/step" % rec["hip_kernel_us_per_step"],
              "library GEMM kernels:", list(rec["library_gemm_kernels"]))
        for k, r in rec["kernels"].items():
            print("   %-36s x%-4.1f %8.1f us  %s" % (k, r["calls_per_step"], r["avg_us"],
                                                   ("%.2f of 8 TB/s" % r["fraction_of_8TBps"]) if "fraction_of_8TBps" in r else ""),
                  ("measured/alg %.2f" % r["measured_over_algorithmic"]) if "measured_over_algorithmic" in r else "")


if __name__ == "__main__":
    main()
