"""Builds nabladft_amd/libnablaq.so (HIP, gfx950) in-tree with plain hipcc.

    python -m nabladft_amd.build [--force]

One object per .hip file (rebuilt only when the source or a header is newer), then one link.
hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the
GPU box with the snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libnablaq.so")
SOURCES = ["graph.hip", "gemm.hip", "gemm_bf16.hip", "edge.hip", "molpair.hip", "updfuse.hip", "node.hip", "schnet.hip", "hblock.hip", "so3.hip", "qhnet.hip", "qhgen.hip", "gemnet_graph.hip", "gemnet.hip", "escn.hip", "equiformer.hip", "geobasis.hip", "rccl.hip", "engine.hip"]
# molpair.hip: the SLP vectoriser packs neighbouring scalar f32 operations into v_pk_* beside the matrix instructions, which costs more than it saves there
# (MI355X_MICROARCH.md, "packed f32 VALU beside MFMAs"; measured 6.51 -> 6.18 ms per step, profiles/r06_gwr_mol_variants.txt)
EXTRA = {"molpair.hip": ["-fno-slp-vectorize"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=on", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "lanes.h"), os.path.join(CSRC, "gemm_tile.h"), os.path.join(CSRC, "gemm_split.h"), os.path.join(CSRC, "cg_l4.inc"), os.path.join(os.path.dirname(HERE), "include", "nablaq.h")]
    objs, procs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + EXTRA.get(s, []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            print(f"hipcc failed on {s}:\n{out}", file=sys.stderr)
            failed = True
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if force or procs or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
