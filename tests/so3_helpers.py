"""Test helper: the reference's Clebsch-Gordan provider rebuilt from the committed fixture (tests/golden/phisnet_cg_l4.npz = the l <= 4
part of the data file the reference's ClebschGordan module loads; same permutation rule as phisnet/nn/modules/clebsch_gordan.py:20-25)."""
import os
from itertools import permutations

import numpy as np
import torch

from tests.helpers import GOLDEN


class FixtureCG:
    def __init__(self, device="cpu"):
        raw = np.load(os.path.join(GOLDEN, "phisnet_cg_l4.npz"))
        self.t = {}
        for name in raw.files:
            l123 = tuple(int(v) for v in name.split("_")[1:])
            for a, b, c in permutations((0, 1, 2)):
                key = (l123[a], l123[b], l123[c])
                if key not in self.t:
                    self.t[key] = torch.tensor(raw[name].transpose(a, b, c)).to(device)

    def __call__(self, l1, l2, l3):
        return self.t[(l1, l2, l3)]
