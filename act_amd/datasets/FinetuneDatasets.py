"""The other finetune datasets of the reference, same config keys, on-disk formats and batch tuples:

ModelNetFewShot      (datasets/ModelNetDatasetFewShot.py:28-71): ``DATA_PATH/{way}way_{shot}shot/{fold}.pkl`` = pickle of
                     ``{'train': [(points[N,6], label, _), ...], 'test': [...]}``; ``way`` / ``shot`` / ``fold`` come from the command line
                     (main.py:72-78) and must all be set.
ScanObjectNN         (datasets/ScanObjectNNDataset.py:11-48): ``ROOT/{training,test}_objectdataset.h5`` with datasets 'data' [n,2048,3], 'label' [n].
ScanObjectNN_hardest (:51-88): ``ROOT/{training,test}_objectdataset_augmentedrot_scale75.h5``.

An item is ``(name, 'sample', (points float32 [N,3], label))``; training items are returned in a random point order."""
import os
import pickle

import numpy as np
import torch
import torch.utils.data as data

from .build import DATASETS
from .SyntheticDataset import pc_norm
from ..utils.logger import print_log


def _shuffled(points, train):
    return points[np.random.permutation(points.shape[0])] if train else points


@DATASETS.register_module()
class ModelNetFewShot(data.Dataset):
    def __init__(self, config):
        self.root = config.DATA_PATH
        self.npoints = config.N_POINTS
        self.use_normals = bool(config.USE_NORMALS)
        self.num_category = config.NUM_CATEGORY
        self.subset = config.subset
        self.way, self.shot, self.fold = config.get("way", -1), config.get("shot", -1), config.get("fold", -1)
        if -1 in (self.way, self.shot, self.fold):
            raise RuntimeError("ModelNetFewShot needs --way, --shot and --fold")
        self.pickle_path = os.path.join(self.root, f"{self.way}way_{self.shot}shot", f"{self.fold}.pkl")
        print_log("Load processed data from %s..." % self.pickle_path, logger="ModelNetFewShot")
        with open(self.pickle_path, "rb") as f:
            self.dataset = pickle.load(f)[self.subset]
        print_log("The size of %s data is %d" % (self.subset, len(self.dataset)), logger="ModelNetFewShot")

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        points, label, _ = self.dataset[index]
        points = np.array(points, dtype=np.float32)               # (the reference normalises the stored array in place; a copy here)
        points[:, 0:3] = pc_norm(points[:, 0:3])
        if not self.use_normals:
            points = points[:, 0:3]
        pts = _shuffled(points, self.subset == "train")
        return "ModelNet", "sample", (torch.from_numpy(np.ascontiguousarray(pts)).float(), label)


def _read_h5(path):
    try:
        import h5py
    except ImportError as e:                                      # not in every image
        raise RuntimeError(f"reading {path} needs the h5py package") from e
    with h5py.File(path, "r") as h5:
        return np.array(h5["data"]).astype(np.float32), np.array(h5["label"]).astype(int)


class _ScanObjectNNBase(data.Dataset):
    FILES = {}

    def __init__(self, config, reader=_read_h5, **kwargs):
        super().__init__()
        self.subset = config.subset
        self.root = config.ROOT
        if self.subset not in self.FILES:
            raise NotImplementedError(self.subset)
        self.points, self.labels = reader(os.path.join(self.root, self.FILES[self.subset]))
        print_log(f"Successfully load ScanObjectNN shape of {self.points.shape}", logger="ScanObjectNN")

    def __len__(self):
        return self.points.shape[0]

    def __getitem__(self, idx):
        pts = _shuffled(self.points[idx], self.subset == "train").copy()
        return "ScanObjectNN", "sample", (torch.from_numpy(pts).float(), self.labels[idx])


@DATASETS.register_module()
class ScanObjectNN(_ScanObjectNNBase):
    FILES = {"train": "training_objectdataset.h5", "test": "test_objectdataset.h5"}


@DATASETS.register_module()
class ScanObjectNN_hardest(_ScanObjectNNBase):
    FILES = {"train": "training_objectdataset_augmentedrot_scale75.h5", "test": "test_objectdataset_augmentedrot_scale75.h5"}
