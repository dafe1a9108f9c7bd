"""Two-crop batching of the unsupervised loader.

Behaviour of ``AspectRatioGroupedDatasetTwoCrop`` (coin/data/common.py:4-47): the mapper yields ``(strong_dict, weak_dict)`` pairs of
one image; pairs are collected in two aspect-ratio groups (w > h, and the rest) and a batch ``(strong list, weak list)`` leaves
as soon as one group holds ``batch_size`` pairs -- images of one batch then need little padding.  The groups keep their
partial content across batches, exactly like the reference's instance-level buckets."""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Tuple


class AspectRatioGroupedDatasetTwoCrop:
    def __init__(self, dataset: Iterable[Tuple[Dict, Dict]], batch_size: int):
        self.dataset, self.batch_size = dataset, batch_size
        self._groups: List[Tuple[List[Dict], List[Dict]]] = [([], []), ([], [])]

    def __iter__(self) -> Iterator[Tuple[List[Dict], List[Dict]]]:
        for strong, weak in self.dataset:
            first, second = self._groups[0 if strong["width"] > strong["height"] else 1]
            first.append(strong)
            second.append(weak)
            if len(first) == self.batch_size:
                yield first[:], second[:]
                del first[:]
                del second[:]
