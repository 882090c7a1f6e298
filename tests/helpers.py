"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from oracle import painn_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    fx = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
    c = fx["cfg"]
    cfg = R.PaiNNConfig(hidden_channels=int(c[0]), num_layers=int(c[1]), num_rbf=int(c[2]), cutoff=float(fx["cutoff"]),
                        max_neighbors=int(c[3]), envelope_exponent=int(c[4]), num_elements=int(c[5]),
                        rbf=str(fx["rbf"]) if "rbf" in fx else "gaussian", direct_forces=bool(fx["direct_forces"]) if "direct_forces" in fx else False)
    params = R.make_params(cfg, int(fx["param_seed"]))
    return fx, cfg, params


def rel_err(a, b):
    """max |a-b| / max |b| (array-level relative error, fp64)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0 and b.size == 0:
        return 0.0
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)


def elem_err(a, b, floor=None):
    """Element-wise relative error max_i |a_i - b_i| / max(|b_i|, floor), fp64; floor = rms(b) unless given: an element is measured against its own
    magnitude, but never against less than the typical magnitude of the array (a force component that happens to be ~0 is not held to 1e-5 of itself)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if a.size == 0 and b.size == 0:
        return 0.0
    if floor is None:
        floor = float(np.sqrt((b * b).mean()))
    return float((np.abs(a - b) / np.maximum(np.abs(b), max(floor, 1e-30))).max())


def _report(line):
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_report.txt"), "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass


def assert_close(name, got, ref, tol=1e-5, elem_tol=None):
    """Both measures against a reference array: array-level max|a-b| / max|b| < tol AND element-wise |a_i-b_i| <= elem_tol * max(|b_i|, rms(b))
    (elem_tol defaults to tol).  Appends both numbers to gpurun_out/parity_report.txt."""
    elem_tol = tol if elem_tol is None else elem_tol
    e_arr, e_el = rel_err(got, ref), elem_err(got, ref)
    ok = e_arr < tol and e_el <= elem_tol
    _report(f"{name:60s} array-level {e_arr:.3e} (< {tol:.1e})  element-wise {e_el:.3e} (<= {elem_tol:.1e})  {'ok' if ok else 'ABOVE'}")
    if os.environ.get("NQ_PARITY_REPORT_ONLY") != "1":
        assert ok, (name, e_arr, tol, e_el, elem_tol)
    return e_arr, e_el


def check_grads(fx, grads, tol, label=""):
    """grads: name -> array. Compares against full ('grad:') or sampled ('gidx:/gval:') golden grads."""
    worst = (0.0, None)
    for key in fx:
        if key.startswith("grad:"):
            name = key[5:]
            e = rel_err(np.asarray(grads[name]), fx[key])
        elif key.startswith("gidx:"):
            name = key[5:]
            g = np.asarray(grads[name]).reshape(-1)
            scale = fx["gnorm:" + name] / np.sqrt(g.size)  # rms of the golden tensor
            e = float(np.abs(g[fx[key]].astype(np.float64) - fx["gval:" + name]).max() / max(np.abs(fx["gval:" + name]).max(), scale))
            nrm = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
            e = max(e, abs(nrm - fx["gnorm:" + name]) / fx["gnorm:" + name])
        else:
            continue
        if e > worst[0]:
            worst = (e, name)
        assert e < tol, f"{label} grad {name}: rel err {e:.3e} >= {tol}"
    return worst


# ---- parity against a float64 run of the reference, with the reference's own float32 error as the yardstick ---------------------------------------------
PARITY_FLOOR = 1e-5          # BASELINE.json north_star: "within 1e-5 rel fp32"


def assert_parity(name, got, ref64, ref32=None, floor=PARITY_FLOOR, factor=1.5):
    """err_hip = |got - ref64|max / |ref64|max must be <= max(floor, factor * err_ref) where err_ref is the same measure of the REFERENCE's own float32
    run (ref32) against its float64 run: a float32 pipeline cannot be asked to sit closer to the float64 answer than the reference's float32 pipeline
    does.  The same holds element-wise (elem_err).  Every call appends both measures to gpurun_out/parity_report.txt (kept as evidence under profiles/)."""
    err = rel_err(got, ref64)
    own = rel_err(ref32, ref64) if ref32 is not None else 0.0
    bound = max(floor, factor * own)
    # element-wise: every element against max(|ref_i|, rms(ref)), same yardstick (the reference's own float32 error in the same measure)
    err_el = elem_err(got, ref64)
    own_el = elem_err(ref32, ref64) if ref32 is not None else 0.0
    bound_el = max(floor, factor * own_el)
    ok = err <= bound and err_el <= bound_el
    _report(f"{name:60s} err_hip {err:.3e}  err_ref_fp32 {own:.3e}  bound {bound:.3e}  | element-wise err_hip {err_el:.3e}  err_ref_fp32 {own_el:.3e}  "
            f"bound {bound_el:.3e}  {'ok' if ok else 'ABOVE'}")
    if os.environ.get("NQ_PARITY_REPORT_ONLY") != "1":
        assert err <= bound, (name, err, own, bound)
        assert err_el <= bound_el, (name, "element-wise", err_el, own_el, bound_el)
    return err, own
