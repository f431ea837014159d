"""Synthetic stand-in for ShapeNet55 (reference: datasets/ShapeNet55Dataset.py:9-70).

Yields the reference's batch tuple ``(taxonomy_id, model_id, data[N,3])`` with ``pc_norm`` semantics
(:45-51): centroid removed, divided by the largest radius.  Registered under NAME 'ShapeNet' so the YAML only
swaps the dataset ``_base_`` file; a real ``.npy`` loader is listed under SURVEY 8(f) "next"."""
import numpy as np
import torch
import torch.utils.data as data

from .build import DATASETS


def pc_norm(pc):
    """pc: NxC numpy -> NxC (datasets/ShapeNet55Dataset.py:45-51)."""
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


@DATASETS.register_module()
class ShapeNet(data.Dataset):
    def __init__(self, config):
        self.npoints = config.N_POINTS
        self.subset = config.subset
        self.sample_points_num = config.npoints
        self.num = int(config.get("NUM_SAMPLES", 4096))
        if not config.get("SYNTHETIC", False):
            raise NotImplementedError("ShapeNet55 .npy files are not shipped; use cfgs/dataset_configs/Synthetic.yaml")
        self.seed = 1234 + (0 if self.subset == "train" else 1)

    def __getitem__(self, idx):
        g = np.random.RandomState((self.seed * 1000003 + idx) & 0x7FFFFFFF)
        pts = pc_norm(g.standard_normal((self.sample_points_num, 3))).astype(np.float32)
        return "synthetic", f"{idx:06d}", torch.from_numpy(pts)

    def __len__(self):
        return self.num
