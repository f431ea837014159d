#!/usr/bin/env python3
"""Generates tests/golden/g12_block_mask.npz: the reference's `_mask_center_block` (models/act.py:215-242) on the G1 centres with
injected seed indices.  Run in the build container (needs /root/reference).

    python tests/golden/make_golden_blockmask.py
"""
import os, sys, random
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, save, REF                      # noqa: E402


def main():
    os.chdir(REF)
    install_shims()
    import models.act as act
    center = torch.from_numpy(np.load(os.path.join(HERE, "g1_group.npz"))["center"])        # [4,64,3]
    seeds = [5, 0, 63, 17]
    it = iter(seeds)
    real = random.randint
    random.randint = lambda a, b: next(it)
    try:
        class Stub:                                                  # the method only reads self.mask_ratio
            mask_ratio = 0.8
        mask = act.VisableOnlyMaskTransformer._mask_center_block(Stub(), center)
    finally:
        random.randint = real
    save("g12_block_mask", seed_index=np.array(seeds), mask=mask)


if __name__ == "__main__":
    main()
