"""CPU: Clebsch-Gordan tensors computed from scratch (nabladft_amd/cg.py) against the reference's table (fixture), the generated kernel
table, and the parameter surface of the SO(3) mixing mirrors."""
import os
import re

import numpy as np
import pytest
import torch

from nabladft_amd import cg
from tests.helpers import GOLDEN
from tests.so3_helpers import FixtureCG


def test_canonical_tensors_equal_reference_table_up_to_sign():
    table = FixtureCG()
    signs = cg.path_signs(lambda a, b, c: table(a, b, c).numpy(), cg.ALL_PATHS)
    assert len(signs) == 65 and set(signs) == {1.0, -1.0}
    nnz = 0
    for (l1, l2, L), s in zip(cg.ALL_PATHS, signs):
        T = cg.canonical(l1, l2, L)
        assert np.abs(s * T - table(l1, l2, L).numpy()).max() < 1e-12
        nnz += int((T != 0).sum())
    assert nnz == 2052                                       # 2052 non-zeros over the 65 paths
    bad = lambda a, b, c: 2.0 * table(a, b, c).numpy()
    with pytest.raises(ValueError):
        cg.path_signs(bad, [(1, 1, 2)])


def test_generated_kernel_table_matches_cg_module():
    inc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nabladft_amd", "csrc", "cg_l4.inc")).read()
    inc = "\n".join(ln for ln in inc.splitlines() if not ln.startswith("//"))
    begins = re.findall(r"CG_PATH_BEGIN\((\d+), (\d+), (\d+), (\d+)\)", inc)
    assert [(int(b), int(c), int(d)) for _, b, c, d in begins] == cg.ALL_PATHS
    body = inc.split("CG_PATH_BEGIN")[1:]
    for (pid, l1, l2, L), chunk in zip(begins, body):
        T = np.zeros_like(cg.canonical(int(l1), int(l2), int(L)))
        for ia, ib, Mi, v in re.findall(r"CG_NZ\((\d+), (\d+), (\d+), ([-0-9.e+]+)f\)", chunk):
            T[int(ia) - int(l1) ** 2, int(ib) - int(l2) ** 2, int(Mi)] = float(v)
        assert np.abs(T - cg.canonical(int(l1), int(l2), int(L))).max() < 1e-9


def test_mixing_mirrors_have_the_reference_parameter_surface():
    from nabladft_amd import so3
    fx = np.load(os.path.join(GOLDEN, "phisnet_mixing.npz"))
    table = FixtureCG()
    pm = so3.PairMixing(4, 4, 4, 16, 64, table)
    ref_names = sorted(k.split(":p:")[1] for k in fx.files if k.startswith("pm444:p:"))
    assert sorted(n for n, _ in pm.named_parameters()) == ref_names and len(ref_names) == 65
    assert all(tuple(p.shape) == fx["pm444:p:" + n].shape for n, p in pm.named_parameters())
    sm = so3.SelfMixing(4, 4, 64, table)
    ref_names = sorted(k.split(":p:")[1] for k in fx.files if k.startswith("sm44:p:"))
    assert sorted(n for n, _ in sm.named_parameters()) == ref_names
    assert not [k for k in sm.state_dict() if k.startswith("_")]         # helper buffers are not part of the checkpoint surface
    with pytest.raises(RuntimeError):                                     # no CPU path
        sm([torch.zeros(1, 2, 2 * l + 1, 64) for l in range(5)])
    with pytest.raises(NotImplementedError):
        so3.PairMixing(5, 4, 4, 8, 64, table)
