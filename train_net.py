#!/usr/bin/env python3
"""Launcher with the reference's command-line surface for the hot path (train_net.py + coin/utils/util.py:151-184):

    python train_net.py --config-file configs/coin/PRETRAINS/CLIPDET_synthetic.yaml [--num-gpus N] [KEY VALUE ...]
    python train_net.py --config-file configs/coin/GDINO/foggy_synthetic.yaml SOLVER.MAX_ITER 100

Same flags (``--config-file``, ``--resume``, ``--eval-only``, ``--num-gpus``, ``--num-machines``, ``--machine-rank``, ``--dist-url``,
``--info``, ``--test_model_role``, trailing ``KEY VALUE`` overrides) and the same dispatch on ``cfg.CLOUD.Trainer``.  Only the
trainers of the adaptation-training hot path exist here: ``PRETRAIN`` -> PRETrainer, ``CoinTrainer`` -> CoinTrainer; the collectors
(GDINO / GLIP / CLIP), ORACLE and evaluation-only modes are outside the scope of this build and say so.  With ``AMD.SYNTHETIC.ENABLED``
off the datasets named in ``DATASETS.TRAIN_UNLABEL`` / ``DATASETS.TEST`` are read from ``$DETECTRON2_DATASETS`` (VOC layout).
One process per GPU: with ``--num-gpus N > 1`` the script re-launches itself through ``torch.distributed.run`` (RCCL over xGMI).
"""
from __future__ import annotations

import argparse
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts: see coin_amd/__init__.py
import subprocess
import sys


def default_argument_parser():
    p = argparse.ArgumentParser(description="COIN adaptation-training hot path on MI355X")
    p.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    p.add_argument("--resume", action="store_true", help="MODEL.WEIGHTS is one CoinTrainer checkpoint to continue from")
    p.add_argument("--eval-only", action="store_true", help="perform evaluation only (outside the hot path)")
    p.add_argument("--num-gpus", type=int, default=1, help="number of gpus *per machine*")
    p.add_argument("--num-machines", type=int, default=1, help="total number of machines")
    p.add_argument("--machine-rank", type=int, default=0, help="the rank of this machine (unique per machine)")
    p.add_argument("--dist-url", default="tcp://127.0.0.1:29500", help="rendezvous URL (host:port are used for torch.distributed.run)")
    p.add_argument("--info", default="")
    p.add_argument("--test_model_role", default="targetdet")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE pairs overriding config options")
    return p


def setup(args):
    from coin_amd.config import get_cfg

    cfg = get_cfg()
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(list(args.opts or []))
    return cfg


def main(args):
    from coin_amd.hostenv import apply_rank_affinity, cap_torch_threads

    # a disjoint, NUMA-local core set + thread caps per rank, before the HIP runtime starts (a CPU-device run keeps its host pools)
    affinity = apply_rank_affinity(cpu_compute=any(str(o).lower() == "cpu" for o in (getattr(args, "opts", None) or [])))
    import torch
    import torch.distributed as dist

    cap_torch_threads(affinity)

    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        args.opts = list(args.opts or []) + ["MODEL.DEVICE", f"cuda:{local_rank}"]
    cfg = setup(args)
    if args.eval_only:
        raise SystemExit("--eval-only needs the evaluation stack, which is outside this build's scope")
    from coin_amd.engine import CoinTrainer, PRETrainer

    trainers = {"PRETRAIN": PRETrainer, "CoinTrainer": CoinTrainer}
    if cfg.CLOUD.Trainer not in trainers:
        raise SystemExit(f"CLOUD.Trainer={cfg.CLOUD.Trainer!r}: only {sorted(trainers)} are on the adaptation-training hot path")
    os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
    if args.info:
        with open(os.path.join(cfg.OUTPUT_DIR, "note.txt"), "a", encoding="utf-8") as f:
            f.write(args.info)
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(cfg.SEED + (dist.get_rank() if dist.is_initialized() else 0))  # util.py:89-90: per-rank seed
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    if cfg.AMD.SYNTHETIC.ENABLED:
        trainer = trainers[cfg.CLOUD.Trainer](cfg)
    else:
        # real data (coin/data/build.py:102-176): the unlabelled target set by name -> VOC dataset dicts -> sampler -> image decode threads
        # -> strong / weak views on the GPU -> two-crop batches; the cached teacher results arrive with MODEL.WEIGHTS (resume_or_load)
        import os.path as osp

        from coin_amd.data import LazyTestSet, build_detection_unsupervised_train_loader
        from coin_amd.data.catalog import SPLITS, dataset_root, get_detection_dataset_dicts, thing_classes
        from coin_amd.evaluation import PascalVOCEvaluator

        if not cfg.AMD.CLASS_NAMES:  # the dataset's thing classes (MetadataCatalog.get(name).thing_classes in the reference)
            cfg.AMD.CLASS_NAMES = list(thing_classes(cfg.DATASETS.TRAIN_UNLABEL[0]))
        loader = build_detection_unsupervised_train_loader(cfg, get_detection_dataset_dicts(cfg.DATASETS.TRAIN_UNLABEL), rank=rank, world_size=world)
        trainer = trainers[cfg.CLOUD.Trainer](cfg, data_loader=loader)
        if cfg.DATASETS.TEST and cfg.TEST.EVAL_PERIOD >= 0:
            name = cfg.DATASETS.TEST[0]
            dirname, split = osp.join(dataset_root(), SPLITS[name][0]), SPLITS[name][1]
            # lazily mapped, sharded over the ranks; the evaluator gathers the predictions (coin/data/build.py:28-53 + inference_on_dataset)
            trainer.set_evaluation(LazyTestSet(cfg, get_detection_dataset_dicts([name]), rank=rank, world_size=world),
                                   lambda: PascalVOCEvaluator(dirname, split, thing_classes(name), year=2012))
    if hasattr(trainer, "resume_or_load"):
        trainer.resume_or_load(resume=args.resume)
    trainer.train()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    a = default_argument_parser().parse_args()
    if a.num_gpus * a.num_machines > 1 and "RANK" not in os.environ:
        host, _, port = a.dist_url.replace("tcp://", "").partition(":")
        cmd = [sys.executable, "-m", "torch.distributed.run", f"--nnodes={a.num_machines}", f"--node-rank={a.machine_rank}",
               f"--nproc-per-node={a.num_gpus}", "--master-addr", host or "127.0.0.1", "--master-port", port or "29500"] + sys.argv
        raise SystemExit(subprocess.call(cmd))  # children are started before anything touches the GPU in this process
    main(a)
