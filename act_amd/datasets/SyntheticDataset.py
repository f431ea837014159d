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


@DATASETS.register_module()
class ModelNet(data.Dataset):
    """Synthetic stand-in for ModelNet40 (reference: datasets/ModelNetDataset.py:142-149): yields
    ``('ModelNet', 'sample', (points[N_POINTS,3] float32, label int))``.  Class c is an anisotropic gaussian whose axis
    scales and orientation are a fixed function of c, so a classifier can actually learn the labels."""

    def __init__(self, config):
        self.npoints = config.N_POINTS
        self.num_category = config.NUM_CATEGORY
        self.subset = config.subset
        self.num = int(config.get("NUM_SAMPLES", 512))
        if not config.get("SYNTHETIC", False):
            raise NotImplementedError("ModelNet40 files are not shipped; use cfgs/dataset_configs/SyntheticModelNet40.yaml")
        self.seed = 4321 + (0 if self.subset == "train" else 1)

    def __getitem__(self, index):
        g = np.random.RandomState((self.seed * 1000003 + index) & 0x7FFFFFFF)
        label = int(g.randint(self.num_category))
        c = np.random.RandomState(977 + label)
        scales = 0.25 + 1.5 * c.rand(3)
        ang = 2 * np.pi * c.rand()
        rot = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]])
        pts = (g.standard_normal((self.npoints, 3)) * scales) @ rot.T
        pts = pc_norm(pts).astype(np.float32)
        return 'ModelNet', 'sample', (torch.from_numpy(pts), label)

    def __len__(self):
        return self.num
