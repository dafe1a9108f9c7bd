"""Pascal-VOC formatted datasets -> detectron2-style dataset dicts (first piece).  Host code only.

Behaviour of coin/data/datasets/pascal_voc.py:25-83 (``load_voc_instances``): one dict per id of ``ImageSets/Main/<split>.txt`` with
``file_name`` (``JPEGImages/<id>.<img_format>``), ``image_id``, ``height``/``width`` from ``<size>``, and ``annotations`` = one entry per
``<object>`` whose class is known: ``category_id``, ``bbox`` = (xmin - 1, ymin - 1, xmax, ymax) as floats (1-based inclusive pixel
indices -> 0-based half-open coordinates), ``bbox_mode`` XYXY_ABS.  "difficult" objects are kept; objects of classes outside
``class_names`` are dropped.  Pinned against the reference's loader on a synthetic tree (tests/golden/voc_dataset.json).
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Sequence

XYXY_ABS = 0  # detectron2.structures.BoxMode.XYXY_ABS


def load_voc_instances(dirname: str, split: str, class_names: Sequence[str], img_format: str = "jpg") -> List[Dict]:
    with open(os.path.join(dirname, "ImageSets", "Main", split + ".txt")) as f:
        file_ids = f.read().split()
    index = {name: i for i, name in enumerate(class_names)}
    out = []
    for fid in file_ids:
        root = ET.parse(os.path.join(dirname, "Annotations", fid + ".xml"))
        annos = []
        for obj in root.findall("object"):
            cid = index.get(obj.find("name").text)
            if cid is None:
                continue
            bb = obj.find("bndbox")
            x0, y0, x1, y1 = (float(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax"))
            annos.append({"category_id": cid, "bbox": [x0 - 1.0, y0 - 1.0, x1, y1], "bbox_mode": XYXY_ABS})
        out.append({"file_name": os.path.join(dirname, "JPEGImages", fid + "." + img_format), "image_id": fid,
                    "height": int(root.findall("./size/height")[0].text), "width": int(root.findall("./size/width")[0].text),
                    "annotations": annos})
    return out
