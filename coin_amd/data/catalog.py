"""Names of the reference's built-in Pascal-VOC formatted datasets -> dataset dicts.

Behaviour of ``register_all_pascal_voc`` (coin/data/datasets/builtin.py:121-175): every name maps to a directory under
``$DETECTRON2_DATASETS`` (default ``datasets``), an ``ImageSets/Main/<split>.txt`` list, an image suffix and one of six class
tables; the evaluator type is ``VOCeval`` for all of them.  This is configuration data (the names appear in the reference's YAML
files: ``DATASETS.TRAIN_UNLABEL`` / ``DATASETS.TEST``), kept as a table."""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

from .voc import load_voc_instances

CLASSES = {
    20: ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog", "horse", "motorbike",
         "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"),
    8: ("truck", "car", "rider", "person", "train", "motorcycle", "bicycle", "bus"),
    7: ("person", "rider", "car", "truck", "bus", "motorcycle", "bicycle"),
    6: ("bicycle", "bird", "car", "cat", "dog", "person"),
    3: ("car", "motorbike", "person"),
    1: ("car",),
}

_CITY = "CityScapes_FoggyCityScapes"
# name: (directory, split, number of classes, image suffix)
SPLITS: Dict[str, Tuple[str, str, int, str]] = {
    "citytrain": (_CITY, "train_city", 8, "png"), "cityval": (_CITY, "val_city", 8, "png"),
    "foggytrain": (_CITY, "train_foggy", 8, "png"), "foggyval": (_CITY, "val_foggy", 8, "png"),
    "foggytrain_0.02": (_CITY, "train_foggy_0.02", 8, "png"), "foggyval_0.02": (_CITY, "val_foggy_0.02", 8, "png"),
    "citytrain_car": (_CITY, "train_city_car", 1, "png"), "cityval_car": (_CITY, "val_city_car", 1, "png"),
    "cliparttrain": ("clipart", "all", 20, "jpg"), "clipartval": ("clipart", "all", 20, "jpg"),
    "KITTItrainval": ("KITTI", "train_car", 1, "png"),
    "SIMtrainval_car": ("SIM", "train_car", 1, "jpg"), "SIMtrainval": ("SIM", "train", 3, "jpg"),
    "BDD100Ktrain": ("BDD100K_voc", "train_object", 7, "jpg"), "BDD100Kval": ("BDD100K_voc", "val_object", 7, "jpg"),
}


def dataset_root() -> str:
    return os.getenv("DETECTRON2_DATASETS", "datasets")


def thing_classes(name: str) -> Tuple[str, ...]:
    return CLASSES[SPLITS[name][2]]


def get_detection_dataset_dicts(names: Sequence[str], root: str = None) -> List[Dict]:
    """detectron2 ``get_detection_dataset_dicts`` for these names: the concatenation of their dataset dicts (no filtering: the
    unlabelled target sets are used with ``filter_empty=False``, coin/data/build.py:103-105)."""
    root = dataset_root() if root is None else root
    out: List[Dict] = []
    for name in names:
        if name not in SPLITS:
            raise KeyError(f"Dataset '{name}' is not registered! Available datasets are: {', '.join(sorted(SPLITS))}")
        dirname, split, ncls, suffix = SPLITS[name]
        out.extend(load_voc_instances(os.path.join(root, dirname), split, CLASSES[ncls], img_format=suffix))
    return out
