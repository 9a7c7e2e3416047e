"""TEST INFRASTRUCTURE ONLY — golden vectors for aa_to_rotmat from the reference's OWN function
(tokenhmr/lib/utils/geometry.py:5-44, imported in place).   python oracle/gen_golden_geometry.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import, tokenhmr_oracle as O  # noqa: E402


def main():
    ns = ref_import.load()
    g = torch.Generator().manual_seed(77)
    th = torch.cat([2.5 * torch.randn(61, 3, generator=g), torch.zeros(1, 3), 1e-6 * torch.randn(2, 3, generator=g)], 0)
    ref = ns.geometry.aa_to_rotmat(th)
    d = (ref - O.aa_to_rotmat(th)).abs().max().item()
    print("oracle vs reference aa_to_rotmat: max|diff| =", d)
    assert d == 0.0
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "geometry_small.npz"), theta=th.numpy(), rotmat=ref.numpy())


if __name__ == "__main__":
    main()
