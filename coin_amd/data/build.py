"""The unsupervised two-view train loader.

Behaviour of ``build_detection_unsupervised_train_loader`` / ``build_unsupervised_batch_data_loader`` (coin/data/build.py:102-176):
dataset dicts -> infinite shuffled index stream sharded by rank (detectron2 ``TrainingSampler``) -> two-view mapper ->
aspect-ratio grouped batches of ``IMG_PER_BATCH_UNLABEL / world_size`` (strong list, weak list).

The reference maps inside ``DATALOADER.NUM_WORKERS`` worker processes because its mapper is CPU work (Pillow).  Here the pixel
work runs on the GPU (``coin_amd.data.dataset_mapper``), so only the file read + image DECODE stays on the host: a small pool of
threads decodes ahead (Pillow releases the GIL while decoding) and the mapper consumes the decoded arrays in stream order on
the caller's thread -- the random draws therefore happen in one deterministic order whatever the number of decode threads."""
from __future__ import annotations

import collections
import itertools
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence

import torch

from .common import AspectRatioGroupedDatasetTwoCrop
from .dataset_mapper import DatasetMapperUnsupervised, TESTMapper, read_image


class TrainingSampler:
    """detectron2 ``TrainingSampler``: an infinite stream of indices -- shuffled permutations of range(size) drawn from one seeded
    generator shared by all ranks -- of which rank r takes elements r, r + world_size, ..."""

    def __init__(self, size: int, shuffle: bool = True, seed: int = 0, rank: int = 0, world_size: int = 1):
        assert size > 0
        self.size, self.shuffle, self.seed, self.rank, self.world_size = size, shuffle, int(seed), rank, world_size

    def _stream(self) -> Iterator[int]:
        g = torch.Generator()
        g.manual_seed(self.seed)
        while True:
            yield from (torch.randperm(self.size, generator=g) if self.shuffle else torch.arange(self.size)).tolist()

    def __iter__(self) -> Iterator[int]:
        yield from itertools.islice(self._stream(), self.rank, None, self.world_size)


class _MappedStream:
    """dataset dicts in sampler order -> mapper outputs; `decode_threads` images are decoded ahead of the consumer."""

    def __init__(self, dataset_dicts: Sequence[Dict], sampler, mapper: DatasetMapperUnsupervised, decode_threads: int):
        self.dicts, self.sampler, self.mapper, self.threads = dataset_dicts, sampler, mapper, max(int(decode_threads), 0)

    def __iter__(self):
        if self.threads == 0:
            for i in self.sampler:
                yield self.mapper(self.dicts[i])
            return
        pending = collections.deque()
        with ThreadPoolExecutor(self.threads) as pool:
            for i in self.sampler:
                d = self.dicts[i]
                pending.append((d, pool.submit(read_image, d["file_name"], self.mapper.img_format)))
                if len(pending) > self.threads:
                    d0, fut = pending.popleft()
                    yield self.mapper(d0, image=fut.result())
            while pending:  # a finite sampler: hand out what is still being decoded
                d0, fut = pending.popleft()
                yield self.mapper(d0, image=fut.result())


def build_unsupervised_batch_data_loader(dataset_dicts: Sequence[Dict], sampler, mapper, total_batch_size_unlabel: int, *, world_size: int = 1,
                                         aspect_ratio_grouping: bool = True, num_workers: int = 0):
    assert total_batch_size_unlabel > 0 and total_batch_size_unlabel % world_size == 0, \
        "Total unlabel batch size ({}) must be divisible by the number of gpus ({}).".format(total_batch_size_unlabel, world_size)
    if not aspect_ratio_grouping:
        raise NotImplementedError("ASPECT_RATIO_GROUPING = False is not supported yet")
    return AspectRatioGroupedDatasetTwoCrop(_MappedStream(dataset_dicts, sampler, mapper, num_workers), total_batch_size_unlabel // world_size)


def build_detection_unsupervised_train_loader(cfg, dataset_dicts: Sequence[Dict], mapper: Optional[DatasetMapperUnsupervised] = None,
                                              rank: int = 0, world_size: int = 1):
    """`dataset_dicts`: detectron2-style dicts of the unlabelled target set (e.g. coin_amd.data.voc.load_voc_instances)."""
    if cfg.DATALOADER.SAMPLER_TRAIN != "TrainingSampler":
        raise NotImplementedError("{} not yet supported.".format(cfg.DATALOADER.SAMPLER_TRAIN))
    mapper = mapper if mapper is not None else DatasetMapperUnsupervised(cfg, True)
    sampler = TrainingSampler(len(dataset_dicts), seed=cfg.SEED if cfg.SEED >= 0 else 0, rank=rank, world_size=world_size)
    return build_unsupervised_batch_data_loader(dataset_dicts, sampler, mapper, cfg.SOLVER.IMG_PER_BATCH_UNLABEL, world_size=world_size,
                                                aspect_ratio_grouping=cfg.DATALOADER.ASPECT_RATIO_GROUPING, num_workers=cfg.DATALOADER.NUM_WORKERS)


class InferenceSampler:
    """detectron2 ``InferenceSampler``: range(size) cut into contiguous shards, the first size % world_size ranks get one more."""

    def __init__(self, size: int, rank: int = 0, world_size: int = 1):
        shard, left = size // world_size, size % world_size
        sizes = [shard + int(r < left) for r in range(world_size)]
        begin = sum(sizes[:rank])
        self._range = range(begin, min(begin + sizes[rank], size))

    def __iter__(self):
        return iter(self._range)

    def __len__(self):
        return len(self._range)


class LazyTestSet:
    """This rank's shard of the test set (InferenceSampler order) as a re-iterable: an image is decoded / resized / uploaded when the
    evaluation reaches it and released afterwards -- nothing of the test set stays on the device between evaluations (a 10k-image
    set kept mapped was ~20 GB of HBM per rank and minutes of start-up; round-2 ADVICE)."""

    def __init__(self, cfg, dataset_dicts: Sequence[Dict], mapper=None, rank: int = 0, world_size: int = 1):
        self.dicts, self.mapper = list(dataset_dicts), mapper if mapper is not None else TESTMapper(cfg)
        self.indices = list(InferenceSampler(len(self.dicts), rank, world_size))

    def __len__(self):
        return len(self.indices)

    def __iter__(self):
        for i in self.indices:
            yield self.mapper(self.dicts[i])


def build_detection_test_loader(cfg, dataset_dicts: Sequence[Dict], mapper=None, rank: int = 0, world_size: int = 1):
    """coin/data/build.py:28-53: every image of the test set once, in order, sharded over the ranks, batch size 1
    (`BASE_Trainer.test` consumes the batches)."""
    mapper = mapper if mapper is not None else TESTMapper(cfg)
    for i in InferenceSampler(len(dataset_dicts), rank, world_size):
        yield [mapper(dataset_dicts[i])]
