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


def test_packed_list_behaves_like_the_reference_lists():
    """so3.PackedList: list-of-orders views of one packed tensor (host logic only, CPU tensors)."""
    import torch
    from nabladft_amd.so3 import PackedList, _pack, _unpack
    order, F, rows = 2, 4, 5
    packed = torch.arange(rows * 9 * F, dtype=torch.float32).view(rows, 9, F)
    xs = _unpack(packed, order, (1, rows), F)
    assert isinstance(xs, PackedList) and len(xs) == 3
    assert [tuple(x.shape) for x in xs] == [(1, rows, 1, F), (1, rows, 3, F), (1, rows, 5, F)]
    assert torch.equal(xs[1][0, :, 0, :], packed[:, 1, :]) and torch.equal(xs[-1][0], packed[:, 4:9, :])
    assert xs[0].data_ptr() == packed.data_ptr()                                   # views, not copies
    p2, lead = _pack(xs, order, F)
    assert p2 is packed and lead == (1, rows)                                      # the packed tensor is handed on as is
    ys = list(xs)                                                                  # the reference's ``list(xs)`` idiom: a plain list of the views
    assert type(ys) is list and len(ys) == 3
    assert len(xs + xs) == 6 and type(xs + xs) is list                             # list concatenation
    xs[0] = xs[0] * 2                                                              # replacing an element drops the shortcut, the list stays usable
    assert xs.packed is None and torch.equal(xs[0], packed[:, :1, :].view(1, rows, 1, F) * 2)
    p3, _ = _pack(xs, order, F)
    assert torch.equal(p3[:, 0, :], packed[:, 0, :] * 2) and torch.equal(p3[:, 1:, :], packed[:, 1:, :])
    # a different order / width is never short-cut
    zs = _unpack(packed, order, (1, rows), F)
    assert _pack(zs[:2], 1, F)[0].shape == (rows, 4, F)
