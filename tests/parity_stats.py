"""What "the HIP rollouts agree with the oracle" means for a whole sample set (all K, not a spot check): per-sample relative
cost error, and - the quantity the controller consumes - how much softmax WEIGHT the disagreeing samples carry and how far the
weighted-mean update moves when the oracle's costs replace the kernel's (reference mppi_torch weighting: w = exp(-(S - min S) /
lambda), call site mppiisaac/planner/mppi_isaac.py:84)."""
import numpy as np


def agreement(S, So, lam, du=None):
    """S: kernel costs [K] (fp32), So: oracle costs [K] (fp64), du: effective perturbations [H][nu][K] (optional).
    Returns a dict: fractions within 1e-4 / 1e-3 / 1e-2, median, max, the weight mass of the samples beyond 1e-3 (under the
    oracle's weights and under the kernel's: the larger of the two) and, with du, the largest change of the nominal update
    sum_k w_k du_k / eta when the oracle's weights replace the kernel's."""
    S, So = np.asarray(S, np.float64), np.asarray(So, np.float64)
    ok = np.isfinite(S) & np.isfinite(So)
    rel = np.where(ok, np.abs(S - So) / np.maximum(np.abs(So), 1e-300), np.inf)
    w_o = np.where(np.isfinite(So), np.exp(-(So - So[np.isfinite(So)].min()) / lam), 0.0)
    w_k = np.where(np.isfinite(S), np.exp(-(S - S[np.isfinite(S)].min()) / lam), 0.0)
    out = rel > 1e-3
    r = {"n": int(S.size), "within_1e-4": float(np.mean(rel <= 1e-4)), "within_1e-3": float(np.mean(rel <= 1e-3)),
         "within_1e-2": float(np.mean(rel <= 1e-2)), "median": float(np.median(rel)), "max": float(rel.max()),
         "n_outside_1e-3": int(out.sum()),
         "weight_mass_outside_1e-3": float(max(w_o[out].sum() / w_o.sum(), w_k[out].sum() / w_k.sum())),
         "eta_rel_err": float(abs(w_k.sum() - w_o.sum()) / w_o.sum())}
    if du is not None:
        d = np.asarray(du, np.float64)
        r["update_max_abs_diff"] = float(np.abs((d * w_k).sum(-1) / w_k.sum() - (d * w_o).sum(-1) / w_o.sum()).max())
    return r


def fmt(tag, r):
    s = (f"{tag}: all {r['n']} samples vs fp64 oracle: median {r['median']:.1e} within 1e-4 {r['within_1e-4']:.4f} 1e-3 {r['within_1e-3']:.4f} "
         f"1e-2 {r['within_1e-2']:.4f} max {r['max']:.1e} | {r['n_outside_1e-3']} beyond 1e-3 carrying {r['weight_mass_outside_1e-3']:.1e} of eta "
         f"| eta rel err {r['eta_rel_err']:.1e}")
    if "update_max_abs_diff" in r:
        s += f" | nominal update moves {r['update_max_abs_diff']:.1e}"
    return s
