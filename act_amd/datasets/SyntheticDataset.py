"""ShapeNet55 / ModelNet40 datasets (reference: datasets/ShapeNet55Dataset.py:9-70, datasets/ModelNetDataset.py:52-149,
datasets/io.py) with the reference's config keys, on-disk formats and batch tuples, plus a synthetic mode (``SYNTHETIC: true``
in the dataset YAML) for the benchmark / tests: the real files are not shipped and there is no network.

ShapeNet item   : ``(taxonomy_id, model_id, points[npoints,3] float32)``, ``pc_norm`` semantics (:45-51).
ModelNet item   : ``('ModelNet', 'sample', (points[N_POINTS,3|6] float32, label int))``."""
import os
import pickle

import numpy as np
import torch
import torch.utils.data as data

from .build import DATASETS
from ..utils.logger import print_log


def read_points(path):
    """point file -> ndarray (datasets/io.py): .npy, .txt (comma- or whitespace-separated), .h5 (dataset 'data')."""
    ext = os.path.splitext(path)[1]
    if ext == ".npy":
        return np.load(path)
    if ext == ".txt":
        with open(path) as f:
            first = f.readline()
        return np.loadtxt(path, delimiter="," if "," in first else None)
    if ext == ".h5":
        try:
            import h5py
        except ImportError as e:                                  # not in this image
            raise RuntimeError("reading .h5 point files needs h5py") from e
        with h5py.File(path, "r") as f:
            return f["data"][()]
    raise ValueError(f"Unsupported file extension: {ext}")


def pc_norm(pc):
    """pc: NxC numpy -> NxC (datasets/ShapeNet55Dataset.py:45-51)."""
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


@DATASETS.register_module()
class ShapeNet(data.Dataset):
    """``DATA_PATH/{subset}.txt`` lists ``<taxonomy>-<model>.npy`` files under ``PC_PATH`` (8192-point clouds); an item is a random
    ``npoints``-subset, pc_norm'd.  ``whole: true`` prepends the test list (datasets/ShapeNet55Dataset.py:20-33)."""

    def __init__(self, config):
        self.npoints = config.N_POINTS
        self.subset = config.subset
        self.sample_points_num = config.npoints
        self.synthetic = bool(config.get("SYNTHETIC", False))
        if self.synthetic:
            self.num = int(config.get("NUM_SAMPLES", 4096))
            self.seed = 1234 + (0 if self.subset == "train" else 1)
            return
        self.data_root, self.pc_path = config.DATA_PATH, config.PC_PATH
        list_file = os.path.join(self.data_root, f"{self.subset}.txt")
        print_log(f"[DATASET] sample out {self.sample_points_num} points", logger="ShapeNet-55")
        print_log(f"[DATASET] Open file {list_file}", logger="ShapeNet-55")
        with open(list_file) as f:
            lines = f.readlines()
        if config.get("whole"):
            with open(os.path.join(self.data_root, "test.txt")) as f:
                lines = f.readlines() + lines
        self.file_list = []
        for line in (l.strip() for l in lines):
            if not line:
                continue
            taxonomy_id, rest = line.split("-", 1)
            self.file_list.append({"taxonomy_id": taxonomy_id, "model_id": rest.split(".")[0], "file_path": line})
        print_log(f"[DATASET] {len(self.file_list)} instances were loaded", logger="ShapeNet-55")

    def __getitem__(self, idx):
        if self.synthetic:
            g = np.random.RandomState((self.seed * 1000003 + idx) & 0x7FFFFFFF)
            pts = pc_norm(g.standard_normal((self.sample_points_num, 3))).astype(np.float32)
            return "synthetic", f"{idx:06d}", torch.from_numpy(pts)
        sample = self.file_list[idx]
        pc = read_points(os.path.join(self.pc_path, sample["file_path"])).astype(np.float32)
        pc = pc[np.random.permutation(pc.shape[0])[:self.sample_points_num]]           # random subset without replacement
        return sample["taxonomy_id"], sample["model_id"], torch.from_numpy(pc_norm(pc).astype(np.float32))

    def __len__(self):
        return self.num if self.synthetic else len(self.file_list)


@DATASETS.register_module()
class ModelNet(data.Dataset):
    """Synthetic stand-in for ModelNet40 (reference: datasets/ModelNetDataset.py:142-149): yields
    ``('ModelNet', 'sample', (points[N_POINTS,3] float32, label int))``.  Class c is an anisotropic gaussian whose axis
    scales and orientation are a fixed function of c, so a classifier can actually learn the labels."""

    def __init__(self, config):
        self.npoints = config.N_POINTS
        self.num_category = config.NUM_CATEGORY
        self.subset = config.subset
        self.use_normals = bool(config.get("USE_NORMALS", False))
        self.synthetic = bool(config.get("SYNTHETIC", False))
        if self.synthetic:
            self.num = int(config.get("NUM_SAMPLES", 512))
            self.seed = 4321 + (0 if self.subset == "train" else 1)
            return
        self._load_files(config.DATA_PATH)

    # ---- file-backed mode (datasets/ModelNetDataset.py:52-118): modelnet{10,40}_shape_names.txt, modelnet*_{train,test}.txt,
    # <root>/<shape>/<shape_id>.txt (x,y,z,nx,ny,nz per line), cached as modelnet{C}_{split}_{N}pts_fps.dat (pickle of two lists)
    def _load_files(self, root):
        assert self.subset in ("train", "test")
        c = self.num_category if self.num_category == 10 else 40
        with open(os.path.join(root, f"modelnet{c}_shape_names.txt")) as f:
            names = [l.rstrip() for l in f if l.strip()]
        classes = {n: i for i, n in enumerate(names)}
        with open(os.path.join(root, f"modelnet{c}_{self.subset}.txt")) as f:
            ids = [l.rstrip() for l in f if l.strip()]
        shape_of = ["_".join(x.split("_")[:-1]) for x in ids]
        self.datapath = [(shape_of[i], os.path.join(root, shape_of[i], ids[i]) + ".txt") for i in range(len(ids))]
        print_log("The size of %s data is %d" % (self.subset, len(self.datapath)), logger="ModelNet")
        cache = os.path.join(root, "modelnet%d_%s_%dpts_fps.dat" % (self.num_category, self.subset, self.npoints))
        if os.path.exists(cache):
            print_log("Load processed data from %s..." % cache, logger="ModelNet")
            with open(cache, "rb") as f:
                self.list_of_points, self.list_of_labels = pickle.load(f)
            return
        print_log("Processing data %s (only running in the first time)..." % cache, logger="ModelNet")
        from ..pointnet2_ops import pointnet2_utils                       # HIP farthest-point sampling (needs the GPU)
        if not torch.cuda.is_available():
            raise RuntimeError("building the ModelNet FPS cache runs the HIP farthest-point-sampling kernel and needs a GPU")
        self.list_of_points, self.list_of_labels = [], []
        for shape, path in self.datapath:
            pts = read_points(path).astype(np.float32)
            xyz = torch.from_numpy(np.ascontiguousarray(pts[:, :3])).cuda().unsqueeze(0)
            idx = pointnet2_utils.furthest_point_sample(xyz, min(self.npoints, pts.shape[0]))[0].long().cpu().numpy()
            self.list_of_points.append(pts[idx])          # start index 0 (the reference's host FPS starts at a random point)
            self.list_of_labels.append(np.array([classes[shape]]).astype(np.int32))
        with open(cache, "wb") as f:
            pickle.dump([self.list_of_points, self.list_of_labels], f)

    def __getitem__(self, index):
        if not self.synthetic:
            pts, label = self.list_of_points[index].copy(), int(self.list_of_labels[index][0])
            pts[:, 0:3] = pc_norm(pts[:, 0:3])
            if not self.use_normals:
                pts = pts[:, 0:3]
            if self.subset == "train":
                pts = pts[np.random.permutation(pts.shape[0])]
            return 'ModelNet', 'sample', (torch.from_numpy(np.ascontiguousarray(pts)).float(), label)
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
        return self.num if self.synthetic else len(self.datapath)
