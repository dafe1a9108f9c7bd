"""PRETrainer: the CLIPDET pre-training step (coin/engine/pre_train.py:26-327) on one MI355X per process.

``run_step`` follows pre_train.py:178-211: fetch (strong, weak) views, attach the cached teacher targets
(``set_boxes``), run the detector with ``branch="pre_train"`` on strong + weak views together, sum the loss
dict, backward, SGD step.  Differences that do not change values: bf16 autocast needs no GradScaler (the
reference's fp16 autocast does, pre_train.py:84,199-202); no per-step ``empty_cache()/gc.collect()``
(pre_train.py:210-211); metrics are read back every ``log_period`` steps instead of every step.
Data parallelism = one process per GPU over RCCL, gradients all-reduced in 32 MiB slices overlapped with backward
(coin_amd.parallel.GradReducer); BatchNorm statistics stay per GPU (broadcast_buffers=False in pre_train.py:59-62).
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .. import graphs
from ..data import SyntheticTwoViewLoader
from ..modeling import build_model
from ..solver import build_lr_scheduler, build_optimizer
from .base import BASE_Trainer


class PRETrainer(BASE_Trainer):
    reducer = None  # coin_amd.parallel.GradReducer when world_size > 1

    def __init__(self, cfg, data_loader=None, collect_model=None):
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world_size > 1 else 0
        assert cfg.CLOUD.PRE_TRAIN_NAME == "CLIP", "only the CLIP pre-train flavour exists (pre_train.py:40,55)"
        self.model = build_model(cfg)
        self.model.train()
        self.optimizer = build_optimizer(cfg, self.model, name="all")
        # data parallelism (pre_train.py:59-62): replicas synchronised once, gradients averaged per step (coin_amd.parallel)
        # cfg.AMD.GRAD_ARENA / COIN_GRAD_ARENA=1: the arena without a collective on ONE GPU (measurements: 97.0 vs 98.6 views/s plain)
        force_ddp = os.environ.get("COIN_FORCE_DDP") == "1" and dist.is_available() and dist.is_initialized()  # 1-rank dry run of the collective path
        self.reducer = None
        if self.world_size > 1 or force_ddp or ((cfg.AMD.GRAD_ARENA or os.environ.get("COIN_GRAD_ARENA") == "1") and self.device.type == "cuda"):
            from ..parallel import GradReducer, broadcast_parameters, force_collectives

            force_collectives(force_ddp)
            broadcast_parameters(self.model)
            self.reducer = GradReducer(self.optimizer.params)
        self.scheduler = build_lr_scheduler(cfg, self.optimizer)
        if data_loader is None:
            assert cfg.AMD.SYNTHETIC.ENABLED, "pass data_loader= (coin_amd.data.build_detection_unsupervised_train_loader) when AMD.SYNTHETIC is off"
            per_gpu = cfg.SOLVER.IMG_PER_BATCH_UNLABEL // self.world_size
            assert per_gpu >= 1 and cfg.SOLVER.IMG_PER_BATCH_UNLABEL % self.world_size == 0  # coin/data/build.py:153-157
            data_loader = SyntheticTwoViewLoader(per_gpu, cfg.AMD.SYNTHETIC.HEIGHT, cfg.AMD.SYNTHETIC.WIDTH, len(cfg.AMD.CLASS_NAMES),
                                                 cfg.AMD.SYNTHETIC.BOXES_PER_IMAGE, seed=cfg.SEED + self.rank, device=self.device,
                                                 num_images=max(per_gpu, cfg.AMD.SYNTHETIC.NUM_IMAGES // self.world_size))
            collect_model = data_loader.cache
        self._data_loader_iter = iter(data_loader)
        self._next_batch = None  # one batch of look-ahead: the frozen stages of the next views run during this step
        self.collect_model = collect_model
        self.iter = self.start_iter = 0
        self.max_iter = cfg.SOLVER.MAX_ITER
        self.last_losses: Optional[Dict[str, torch.Tensor]] = None

    def set_boxes(self, unlabel_datas: List[List[Dict]], thresh=None):
        for unlabel_data in unlabel_datas:
            for d in unlabel_data:
                res = self.collect_model(d["file_name"])
                assert res["height"] == d["height"] and res["width"] == d["width"] and res["image_id"] == d["image_id"]
                res = self.preprocess_results(res, tuple(d["image"].shape[1:]), d["random_flip"], thresh=thresh)
                rc = res["RCNN"]
                rc.gt_classes_offline, rc.gt_probs_offline, rc.gt_scores_offline = rc.gt_classes, rc.probs, rc.scores
                for k in ("scores", "gt_classes", "probs"):
                    rc.remove(k)
                rp = res["RPN"]
                rp.remove("scores")
                rp.remove("probs")
                d["RCNN"], d["RPN"] = rc, rp
        return unlabel_datas

    def run_step(self):
        assert self.model.training, "[PTrainer] model was changed to eval mode!"
        thresh = 0.5 if tuple(self.cfg.DATASETS.TRAIN_UNLABEL) == ("cliparttrain",) else None

        def fetch():
            strong, weak = next(self._data_loader_iter)
            strong, weak = self.set_boxes([strong, weak], thresh=thresh)
            strong.extend(weak)
            return strong

        strong = self._next_batch if self._next_batch is not None else fetch()
        self._next_batch = fetch() if getattr(self.model, "overlap_streams", False) and self.device.type == "cuda" else None
        if self._next_batch is not None:
            self.model.set_lookahead(self._next_batch)
        start = self.cfg.CLOUD.PROTOTYPE_UPDATE_START
        update_prototype = start != -1 and self.iter >= start
        record = self.model(strong, branch="pre_train", update_prototype=update_prototype)
        losses = sum(record.values())
        self.optimizer.zero_grad()
        losses.backward()   # with a reducer: 32 MiB gradient slices are all-reduced over RCCL while backward is still running
        self.optimizer.step(inv_loss_scale=self.reducer.finalize() if self.reducer is not None else 1.0)
        self.scheduler.step()
        graphs.step_done()
        self.last_losses = record
        self.iter += 1
        return record

    # ------------------------------------------------------------------ files (pre_train.py:138-146, 172-175, 238-279)
    def save(self, path: str, iteration: Optional[int] = None, load_models: bool = True):
        """Pre-train checkpoint in the reference's layout (DetectionTSCheckpointer(model, optimizer=, scheduler=).save + pre_train.py:138-146):
        {"model", "optimizer", "scheduler", "iteration", "results"[, "load_models": False]} -- what a CoinTrainer run takes as the
        first half of ``MODEL.WEIGHTS`` ("pre_train_CLIP_xxx.pth+GDINO_collect.pth")."""
        from ..checkpoint import optimizer_state, save_file, scheduler_state

        results = self.collect_model.get_results() if hasattr(self.collect_model, "get_results") else None
        blob = {"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}, "optimizer": optimizer_state(self.optimizer),
                "scheduler": scheduler_state(self.scheduler), "iteration": self.iter - 1 if iteration is None else iteration, "results": results,
                "ap_50": dict(self.ap_50 or {})}   # MyPeriodicCheckpointer(..., ap_50=self.ap_50) (pre_train.py:316-319)
        if not load_models:  # pre_train.py:142: the key exists only when False (the collection run's CLIP_-0000001.pth)
            blob["load_models"] = False
        save_file(blob, path)

    def resume_or_load(self, resume: bool = False):
        """pre_train.py:238-279.  ``MODEL.WEIGHTS`` = a pre-train checkpoint or the file written by the collection run: the detector
        weights are loaded in both modes (fvcore Checkpointer.load always loads "model") unless the file says ``load_models: False``
        (then the fresh initialisation is kept, :265-268); ``start_iter`` = stored iteration + 1; the cached teacher results are
        taken from "results"; with ``--resume`` the optimizer and the scheduler are restored too.  Synthetic runs (no weights given)
        keep the generated cache."""
        from ..checkpoint import CloudResults, detector_state_dict, load_file, load_optimizer_state

        if not self.cfg.MODEL.WEIGHTS:
            return
        blob = load_file(self.cfg.MODEL.WEIGHTS)
        is_ckpt = isinstance(blob, dict) and "model" in blob
        if not (is_ckpt and blob.get("load_models", True) is False):
            self.model.load_state_dict(detector_state_dict(blob), strict=False)
        if resume and is_ckpt:
            if blob.get("optimizer") is not None:
                load_optimizer_state(self.optimizer, blob["optimizer"])
            if blob.get("scheduler") is not None:
                self.scheduler.load_state_dict(blob["scheduler"])
        if is_ckpt:
            self.iter = self.start_iter = blob.get("iteration", -1) + 1
            self.ap_50 = dict(blob.get("ap_50") or {}) or None
            if blob.get("results") is not None:
                self.collect_model = CloudResults(blob["results"], device=self.device)
        self._next_batch = None

    def after_step(self):
        """pre_train.py:172-175 + MyPeriodicCheckpointer(file_prefix=CLOUD.PRE_TRAIN_NAME) (hooks.py:60-84): the final model as
        ``pre_train_CLIP_<iter>.pth``, periodic ``CLIP_<iter>.pth`` and ``CLIP_final.pth`` after the last iteration -- independent
        conditions, as in the reference.  `self.iter` has already advanced past the step that just finished."""
        done = self.iter - 1
        if self.rank != 0 or not self.cfg.OUTPUT_DIR:
            return
        period, prefix, out = self.cfg.SOLVER.CHECKPOINT_PERIOD, self.cfg.CLOUD.PRE_TRAIN_NAME, self.cfg.OUTPUT_DIR
        names = []
        if done == self.max_iter - 1:
            names.append("pre_train_{}_{:07d}.pth".format(prefix, done))
        if period > 0 and (done + 1) % period == 0:
            names.append("{}_{:07d}.pth".format(prefix, done))
        if done >= self.max_iter - 1:
            names.append("{}_final.pth".format(prefix))
        if names:
            os.makedirs(out, exist_ok=True)
        for n in names:
            self.save(os.path.join(out, n), iteration=done)

    ap_50: Dict[int, float] = None

    def before_checkpoint(self):
        """test_and_save_results_student (pre_train.py:300-308): evaluate the student when due and remember its AP50 per iteration
        (the dict travels in the checkpoints, hooks.py:60-84)."""
        self.iter -= 1  # hooks see the index of the step just finished
        try:
            if self._eval_due():
                res = self._evaluate(self.model)
                self._last_eval_results_student = res
                if self.ap_50 is None:
                    self.ap_50 = {}
                self.ap_50[self.iter] = res["bbox"]["AP50"]
        finally:
            self.iter += 1

    def train(self):
        for _ in range(self.start_iter, self.max_iter):
            rec = self.run_step()
            self.before_checkpoint()
            self.after_step()
            m = self._write_metrics(rec, self.iter)
            if m is not None and self.rank == 0:
                print(f"iter {self.iter}: " + "  ".join(f"{k} {v:.4f}" for k, v in m.items()), flush=True)
