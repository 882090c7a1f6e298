"""Throughput against conformers per step for the models whose records were only ever taken at 2 / 16 conformers (VERDICT r5 weak #11 iv, item 4): QHNet,
GemNet-OC, eSCN, EquiformerV2 at growing batch sizes until the rate saturates (< 5 % gain for a doubling) or the memory ends; the saturating batch and its
conformer-steps/s are each model's single-GPU figure.  Pure measurement (scripts/bench_<model>.py `run`, no kernel tables): one subprocess per point so that an
out-of-memory point does not take the sweep down.

    python scripts/batch_sweep.py [--out profiles/r06_batch_sweep.json] [--steps 5]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POINT = r"""
import json, sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/scripts")
import bench_%(mod)s as B
dev = torch.device("cuda:0")
r = B.run(molecules=%(n)d, steps=%(steps)d, warmup=2, kernels=False, device=dev)
print("POINT " + json.dumps({"conformers": %(n)d, "ms_per_step": r["ms_per_step"], "value": r["value"], "atoms_per_step": r.get("atoms"),
                             "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))
"""
SWEEPS = {"qhnet": [2, 16, 64, 128, 256], "gemnet": [8, 16, 64, 256, 512], "escn": [8, 16, 64, 256, 512], "equiformer": [8, 16, 64, 256, 512]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "batch_sweep.json"))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--models", default=",".join(SWEEPS))
    args = ap.parse_args()
    out = {}
    for mod in args.models.split(","):
        pts, best = [], None
        for n in SWEEPS[mod]:
            code = POINT % {"root": ROOT, "mod": mod, "n": n, "steps": args.steps}
            try:
                p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=420)
            except subprocess.TimeoutExpired:
                pts.append({"conformers": n, "error": "timeout (420 s)"})
                break
            line = [l for l in p.stdout.splitlines() if l.startswith("POINT ")]
            if p.returncode != 0 or not line:
                err = (p.stderr or "").strip().splitlines()[-1:] or ["failed"]
                pts.append({"conformers": n, "error": err[0][:200]})
                break
            pt = json.loads(line[0][6:])
            pts.append(pt)
            print(mod, pt, flush=True)
            if best is not None and pt["value"] < 1.05 * best["value"]:
                if pt["value"] > best["value"]:
                    best = pt
                break                                       # saturated
            if best is None or pt["value"] > best["value"]:
                best = pt
        out[mod] = {"points": pts, "single_gpu_figure": best,
                    "rule": "first batch whose doubling (x4 here) gains < 5 %, or the largest that ran"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v["single_gpu_figure"] for k, v in out.items()}))


if __name__ == "__main__":
    main()
