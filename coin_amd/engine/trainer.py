"""CoinTrainer: the target-detector (targetDET) distillation step of COIN on one MI355X per process.

``run_step`` follows coin/engine/trainer.py:160-218:

  1. EMA of the student into the offline (CLIP-detector) teacher every OFFLINE_TEACHER_UPDATE_ITER steps once the burn-up
     phase is over (ts_ensemble.py:39-69) -- one ``coin_ema_update`` launch over the whole state dict;
  2. teacher inference on the weak views (``branch='test'``, no grad);
  3. ``match_boxes``: teacher detections + cached cloud-detector results -> (A, B, C) targets for the RoI head and
     (A, None, C) for the RPN (``coin_amd.engine.matching``);
  4. student forward on the strong views, branch ``step_one`` (burn-up) or ``step_two``, with the CKG merge module;
  5. CKG update: ``loss_merge_grad`` (gradient_discrepancy_loss) + ``loss_merge_base`` -> merge optimizer;
  6. student update with the remaining loss terms.

Differences that do not change values: step 5 differentiates only with respect to the CKG parameters
(``backward(inputs=...)``) -- the reference back-propagates this loss through the whole detector and then throws those
gradients away (``optimizer.zero_grad()``, trainer.py:199); bf16 autocast needs no GradScaler; no per-step
``empty_cache()/gc.collect()``; metrics are read back every ``log_period`` steps.
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .. import graphs
from .. import kernels as K
from .. import streams as _streams
from ..data import SyntheticTwoViewLoader
from ..modeling import build_model
from ..modeling.text_encoder import build_merge
from ..solver import build_lr_scheduler, build_optimizer
from .base import BASE_Trainer
from .matching import match_dual_teacher

_MERGE_TERMS = ("loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base")


class CoinTrainer(BASE_Trainer):
    def __init__(self, cfg, data_loader=None, cloud_results=None):
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world_size > 1 else 0
        self.model, self.offline_teacher = build_model(cfg)
        self.merge = build_merge(cfg).to(self.device)
        for p in self.offline_teacher.parameters():
            p.requires_grad = False
        self.model.train()
        self.offline_teacher.train()
        self.optimizer = build_optimizer(cfg, self.model, name="all")
        self.optimizer_merge = build_optimizer(cfg, self.merge, name="all")
        self.ddp_model, self.ddp_merge = self.model, self.merge  # kept for callers of the reference's attribute names
        self.reducer = self.reducer_merge = None
        # trainer.py:66-72: the student AND the CKG module are data parallel (cfg.AMD.GRAD_ARENA: the same arenas on one GPU, no collective)
        if self.world_size > 1 or ((cfg.AMD.GRAD_ARENA or os.environ.get("COIN_GRAD_ARENA") == "1") and self.device.type == "cuda"):
            from ..parallel import GradReducer, broadcast_parameters

            broadcast_parameters(self.model)
            broadcast_parameters(self.merge)
            self.reducer = GradReducer(self.optimizer.params)
            self.reducer_merge = GradReducer(self.optimizer_merge.params)
        self.scheduler = build_lr_scheduler(cfg, self.optimizer)
        self.scheduler_merge = build_lr_scheduler(cfg, self.optimizer_merge)
        if data_loader is None:
            assert cfg.AMD.SYNTHETIC.ENABLED, "pass data_loader= (coin_amd.data.build_detection_unsupervised_train_loader) when AMD.SYNTHETIC is off"
            per_gpu = cfg.SOLVER.IMG_PER_BATCH_UNLABEL // self.world_size
            assert per_gpu >= 1 and cfg.SOLVER.IMG_PER_BATCH_UNLABEL % self.world_size == 0  # coin/data/build.py:153-157
            data_loader = SyntheticTwoViewLoader(per_gpu, cfg.AMD.SYNTHETIC.HEIGHT, cfg.AMD.SYNTHETIC.WIDTH, len(cfg.AMD.CLASS_NAMES),
                                                 cfg.AMD.SYNTHETIC.BOXES_PER_IMAGE, seed=cfg.SEED + self.rank, device=self.device,
                                                 num_images=max(per_gpu, cfg.AMD.SYNTHETIC.NUM_IMAGES // self.world_size))
            cloud_results = data_loader.cache
        self._data_loader_iter = iter(data_loader)
        self.model_CLOUD = cloud_results  # file name -> cached cloud-detector result (gdino_collector.py:51-75)
        self.iter = self.start_iter = 0
        self.max_iter = cfg.SOLVER.MAX_ITER
        self.WEIGHT_FOR_BOX_A = 1.0
        self.last_losses: Optional[Dict[str, torch.Tensor]] = None
        self._ema = None

    # ------------------------------------------------------------------ teacher EMA (ts_ensemble.py:39-69)
    def _build_ema(self, loose):
        """fp32 state-dict entries -> one table for `coin_ema_update` (raw pointers), except the keys in `loose`; integer buffers and the
        loose keys are updated with torch ops on the tensors the modules hold at that moment."""
        t_sd, s_sd = self.offline_teacher.state_dict(), self.model.state_dict()
        missing = [k for k in t_sd if k not in s_sd]
        if missing:
            raise Exception("{} is not found in student model".format(missing[0]))
        fl = [k for k, v in t_sd.items() if v.dtype == torch.float32 and k not in loose]
        owners = {}
        for side, model in (("t", self.offline_teacher), ("s", self.model)):
            for prefix, mod in model.named_modules():
                for name in mod._buffers:
                    owners[(side, (prefix + "." if prefix else "") + name)] = (mod, name)
        # student buffers inside the table: their storage must still be the one the table points at when the kernel runs
        watch = [(k, *owners[("s", k)], s_sd[k].data_ptr()) for k in fl if ("s", k) in owners]
        slow = [k for k, v in t_sd.items() if v.dtype != torch.float32 or k in loose]
        self._ema = (K.EmaTable([t_sd[k] for k in fl], [s_sd[k] for k in fl]), [(owners[("t", k)], owners[("s", k)]) for k in slow], watch, set(loose))

    @torch.no_grad()
    def update_teacher(self, keep_rate: float):
        if self._ema is None:
            self._build_ema(set())
        # A module may REBIND a buffer (`buf.data = new`, the reference's prototype updates, fast_rcnn.py:545-556 -- the old storage
        # stays alive for that step's backward): the table would keep averaging the stale storage.  Such keys leave the table for good.
        moved = [k for k, mod, name, ptr in self._ema[2] if mod._buffers[name].data_ptr() != ptr]
        if moved:
            self._build_ema(self._ema[3] | set(moved))
        table, slow, _, _ = self._ema
        table.update(keep_rate)
        for m in self.offline_teacher.modules():   # the kernel writes through raw pointers: drop results cached on the old prompt vectors
            if hasattr(m, "invalidate_text_cache"):
                m.invalidate_text_cache()
        for (tm, tn), (sm, sn) in slow:  # integer buffers (num_batches_tracked): float arithmetic, then truncation on the copy
            t, s_ = tm._buffers[tn], sm._buffers[sn]
            t.copy_(s_ * (1 - keep_rate) + t * keep_rate)

    # ------------------------------------------------------------------ targets (trainer.py:463-485)
    @torch.no_grad()
    def match_boxes(self, batched_input: List[Dict], offline_results: List[Dict], to_device: bool = True):
        """to_device=False leaves the (A, B, C) targets on the host (`_targets_to_device` moves them on the consumer's stream)."""
        rcnn, rpn = [], []
        thr = self.cfg.CLOUD.MATCHER.IOU_THRESHOLDS
        for data, off in zip(batched_input, offline_results):
            dev = off["instances"].pred_boxes.tensor.device if to_device else None
            online = self.model_CLOUD(data["file_name"])
            net = tuple(data["image"].shape[1:])
            off_i = self.process(off["instances"].to("cpu"), (data["height"], data["width"]), net, "no")
            assert online["height"] == data["height"] and online["width"] == data["width"] and online["image_id"] == data["image_id"]
            online = self.preprocess_results(online, net, data["random_flip"], thresh=None)
            rcnn.append(match_dual_teacher(online, off_i, "RCNN", thr, self.WEIGHT_FOR_BOX_A, device=dev))
            rpn.append(match_dual_teacher(online, off_i, "RPN", thr, self.WEIGHT_FOR_BOX_A, device=dev))
        return rcnn, rpn

    # ------------------------------------------------------------------ one step (trainer.py:160-218)
    def _fetch(self):
        """The next batch and its targets: `next(loader)`, then steps 1-3 of the iteration `self.iter` (EMA of the teacher when due,
        teacher inference on the weak views, matching) -> (strong views, (A, B, C) targets).

        On the GPU (cfg.AMD.TEACHER_STREAM) ALL of it runs on the teacher's own HIP stream -- the loader included: with real
        files the loader uploads the decoded image and queues the coin_aug_* kernels that WRITE the two views
        (DatasetMapperUnsupervised); drawn on the main stream those kernels would sit behind the student's backward while the
        teacher stream read the still-empty buffers (round-2 ADVICE: garbage detections -> silently wrong pseudo-labels; the
        synthetic loader, whose images are resident, hid it).  The batch is produced on the stream that consumes it first; the main
        stream takes the strong views after its `wait_stream(teacher)` in `run_step` (the tensors are marked with
        `record_stream`, as they were allocated from the teacher stream's pool).
        The read-back of the detections waits for the teacher's kernels only -- not for the student's backward that `prepare_next`
        left queued on the main stream -- so the host walks through the matcher while the device is still busy with the student,
        and the teacher's (small, low-occupancy) inference kernels share the GPU with that backward.  The only cross-stream
        dependency is the EMA, which reads the weights the optimizer has just written: the teacher stream waits for the main
        stream when an EMA is due.  The targets stay on the host until `run_step` uploads them on the main stream."""
        return self._fetch_end(self._fetch_begin(self.iter))

    _teacher_mods = None

    def _teacher_mode(self, training: bool):
        """`self.offline_teacher.train(training)` without nn.Module.train()'s recursive Python walk (600 modules, 1.2 ms per call and
        four calls per iteration: the reference brackets every teacher pass with eval() / train(), trainer.py:175-177)."""
        if self._teacher_mods is None:
            self._teacher_mods = list(self.offline_teacher.modules())
        for m in self._teacher_mods:
            object.__setattr__(m, "training", training)

    def _ema_due(self, it: int) -> bool:
        burn = self.cfg.CLOUD.BURN_UP_STEP
        return it >= burn and (it - burn) % self.cfg.CLOUD.OFFLINE_TEACHER_UPDATE_ITER == 0

    def _fetch_begin(self, it: int):
        """First half of `_fetch` for iteration `it`: draw the batch, EMA when due, and ENQUEUE the teacher's inference pass without a
        host round trip (OpenVocabularyRCNN.inference_begin; a detector that has no fixed-shape pass leaves everything to the second
        half).  While the teacher is frozen (no EMA due for `it`: the whole step_one phase, trainer.py:170-172) `run_step` calls this
        for iteration i+1 BEFORE it enqueues the student's step i: the same computation on the same weights, but the teacher's
        device work and the read-back of its detections then hide under the student's step instead of following it."""
        cfg = self.cfg
        ema_due = self._ema_due(it)
        side = main = None
        amd = getattr(cfg, "AMD", None)
        # An EMA-due pass has to follow the optimizer step on the device whatever stream it is on (the EMA reads the new weights), so it
        # loses nothing on the default stream -- and from there its fixed-shape half can be replayed as ONE HIP graph (graph replay
        # from a side stream serialises the device on this runtime).
        use_graph = (ema_due and self.device.type == "cuda" and getattr(amd, "TEACHER_GRAPH", True) and os.environ.get("COIN_TEACHER_GRAPH", "1") != "0"
                     and hasattr(self.offline_teacher, "inference_begin") and not getattr(self.offline_teacher, "graph_failed", False)
                     and self.offline_teacher._static_inference_ok())
        if self.device.type == "cuda" and getattr(amd, "TEACHER_STREAM", True) and not use_graph:
            main = torch.cuda.current_stream(self.device)
            if self._teacher_stream is None:
                self._teacher_stream = _streams.role_stream(self.device, "teacher")
                ema_due_or_first = True
            else:
                ema_due_or_first = ema_due
            side = self._teacher_stream
            if ema_due_or_first:
                side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            strong, weak = next(self._data_loader_iter)
            if ema_due:
                self.update_teacher(cfg.CLOUD.EMA_KEEP_RATE_OFFLINE)
            if hasattr(self.offline_teacher, "inference_begin"):
                self._teacher_mode(False)
                if use_graph:
                    self.offline_teacher.inference_begin(weak, branch="test", graph=True)
                else:
                    self.offline_teacher.inference_begin(weak, branch="test")
                self._teacher_mode(True)
        return strong, weak, side, main

    def _fetch_end(self, begun):
        strong, weak, side, main = begun
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            with torch.no_grad():
                self._teacher_mode(False)
                offline_results = self.offline_teacher(weak, branch="test")   # picks the begun pass up (same `weak` object)
                self._teacher_mode(True)
                targets = self.match_boxes(weak, offline_results, to_device=side is None)
        if side is not None:
            for d in strong:  # consumed by the main stream (after its wait_stream in run_step)
                if torch.is_tensor(d.get("image")) and d["image"].is_cuda:
                    d["image"].record_stream(main)
        return strong, targets

    def _targets_to_device(self, targets):
        """(A, B, C) tuples of `match_boxes(to_device=False)` -> device, on the current (consumer's) stream."""
        rcnn, rpn = targets
        mv = lambda t: tuple(x.to(self.device) if x is not None else None for x in t)
        return [mv(t) for t in rcnn], [mv(t) for t in rpn]

    _pending = None
    _early = None          # (strong, weak, streams) of iteration i+1 whose teacher pass was enqueued ahead of step i
    _teacher_stream = None
    reducer = reducer_merge = None  # coin_amd.parallel.GradReducer when world_size > 1

    def run_step(self):
        cfg = self.cfg
        assert self.model.training, "[PTrainer] model was changed to eval mode!"
        burn = cfg.CLOUD.BURN_UP_STEP
        if self._pending is None:
            if self._early is not None:   # a caller that does not use prepare_next(): finish the pass begun during the last step
                early, self._early = self._early, None
                strong, dual_teacher_instances = self._fetch_end(early)
            else:
                strong, dual_teacher_instances = self._fetch()
        else:
            strong, dual_teacher_instances = self._pending
            self._pending = None
        if self._teacher_stream is not None:
            # the teacher stream's EMA read the student's weights: nothing queued from here on (this step's optimizer in particular)
            # may overtake it.  By now its work is finished in practice (the host waited for the detections), so this costs nothing.
            torch.cuda.current_stream(self.device).wait_stream(self._teacher_stream)
            dual_teacher_instances = self._targets_to_device(dual_teacher_instances)
        if (self._early is None and self.device.type == "cuda" and self.iter + 1 < self.max_iter
                and not self._ema_due(self.iter + 1) and getattr(getattr(cfg, "AMD", None), "TEACHER_PREFETCH", True)):
            # frozen teacher: issue the NEXT iteration's teacher pass now, ahead of this iteration's student step (see _fetch_begin);
            # AFTER the wait above, so that the student's forward does not queue behind it.  It reads the teacher's weights only.
            self._early = self._fetch_begin(self.iter + 1)
        start = cfg.CLOUD.PROTOTYPE_UPDATE_START
        update_prototype = start != -1 and self.iter >= start
        branch = "step_one" if self.iter < burn else "step_two"
        record = self.model(strong, self.merge, dual_teacher_instances, branch=branch, update_prototype=update_prototype)
        self.optimizer.zero_grad()
        self.optimizer_merge.zero_grad()
        has_merge = "loss_merge_a" in record
        if self.world_size > 1:
            # The merge module is data-parallel too (trainer.py:70-72), so its gradient all-reduce must be entered by every rank or by
            # none.  Whether a rank's batch contains B boxes is data dependent (the reference dead-locks when the ranks disagree).  Every
            # rank therefore enters the merge slices every step -- a rank without merge terms contributes zeros -- and the number of ranks
            # WITH merge terms rides in the same all-reduce (GradReducer.flag); the CKG optimizer reads it on the device and skips the
            # update when it is 0 (coin_sgd_step's gate): no collective of its own, no `.item()` (round 3 agreed on a host flag per step).
            self.reducer_merge.set_flag(1.0 if has_merge else 0.0)
            if has_merge:
                record["loss_merge_grad"] = self.model.roi_heads.box_predictor.merge_grad_loss()
                (record["loss_merge_grad"] + record["loss_merge_base"]).backward(inputs=list(self.merge.parameters()), retain_graph=True)
            scale = self.reducer_merge.finalize()
            self.optimizer_merge.step(inv_loss_scale=scale, gate=self.reducer_merge.flag)
        elif has_merge:
            # CKG update (trainer.py:192-197); gradients are formed for the merge parameters only
            record["loss_merge_grad"] = self.model.roi_heads.box_predictor.merge_grad_loss()
            (record["loss_merge_grad"] + record["loss_merge_base"]).backward(inputs=list(self.merge.parameters()), retain_graph=True)
            self.optimizer_merge.step(inv_loss_scale=self.reducer_merge.finalize() if self.reducer_merge is not None else 1.0)
        self.optimizer.zero_grad()
        self.optimizer_merge.zero_grad()
        skip = _MERGE_TERMS if self.iter >= burn else _MERGE_TERMS + ("loss_cls_b",)
        losses = sum(v for k, v in record.items() if k not in skip)
        losses.backward()
        self.optimizer.step(inv_loss_scale=self.reducer.finalize() if self.reducer is not None else 1.0)
        self.scheduler.step()
        self.scheduler_merge.step()
        graphs.step_done()
        self.last_losses = record
        if self.iter >= burn:  # trainer.py:150-157 (after_step): fused A boxes from the next step on
            self.WEIGHT_FOR_BOX_A = 0.5
        self.iter += 1
        return record

    def prepare_next(self):
        """The next iteration begins with the teacher's EMA / inference / matching, which read nothing but the weights just
        updated: doing them right after this iteration's optimizer step is the same computation in the same order.  The
        difference is on the clock: the device is still executing this iteration's backward (enqueued asynchronously) while
        the host goes through the synchronising post-processing of the teacher's detections and the matcher.  Called by
        `train()` AFTER `after_step()`, so a checkpoint written for iteration i holds the teacher as iteration i left it
        (the reference saves in after_step, before the next iteration's EMA: trainer.py:149-172)."""
        if self._pending is None and self.iter < self.max_iter:
            if self._early is not None:
                early, self._early = self._early, None
                self._pending = self._fetch_end(early)
            else:
                self._pending = self._fetch()

    def resume_or_load(self, resume: bool = False):
        """trainer.py:220-262: ``MODEL.WEIGHTS`` = "offline_teacher.pth+cloud_results.pth" (start of adaptation) or one CoinTrainer
        checkpoint.  Synthetic runs (no weights given) keep the random initialisation."""
        from ..checkpoint import load_cointrainer_weights

        if not self.cfg.MODEL.WEIGHTS:
            assert self.cfg.AMD.SYNTHETIC.ENABLED, "pretrain models must be loaded!"
            return
        assert not (resume and "+" in self.cfg.MODEL.WEIGHTS), "resume need only one model."
        load_cointrainer_weights(self, self.cfg.MODEL.WEIGHTS, resume=resume)
        self._pending, self._ema, self._early = None, None, None

    def after_step(self):
        """trainer.py:149-157 + MyPeriodicCheckpointer (hooks.py:60-84): ``burn_up_<iter>.pth`` at the end of the burn-up phase,
        periodic ``model_<iter>.pth``, ``model_final.pth`` after the last iteration (DetectionTSCheckpointer layout) -- independent
        conditions, as in the reference.  `self.iter` has already advanced past the finished step."""
        import os

        from ..checkpoint import save_cointrainer_checkpoint

        done = self.iter - 1
        if self.rank != 0 or not self.cfg.OUTPUT_DIR:
            return
        period = self.cfg.SOLVER.CHECKPOINT_PERIOD
        names = []
        if done == self.cfg.CLOUD.BURN_UP_STEP - 1:
            names.append("burn_up_{:07d}.pth".format(done))
        if period > 0 and (done + 1) % period == 0:
            names.append("model_{:07d}.pth".format(done))
        if done >= self.max_iter - 1:
            names.append("model_final.pth")
        if names:
            os.makedirs(self.cfg.OUTPUT_DIR, exist_ok=True)
        for n in names:
            save_cointrainer_checkpoint(self, os.path.join(self.cfg.OUTPUT_DIR, n), iteration=done)

    ap_50_student: Dict[int, float] = None
    ap_50_offline_teacher: Dict[int, float] = None

    def before_checkpoint(self):
        """build_hooks (trainer.py:296-318): the student is evaluated every TEST.EVAL_PERIOD iterations and after the last one; the
        teacher once at the first evaluation (iteration EVAL_PERIOD - 1), is carried forward while it is frozen (burn-up), and is
        evaluated itself from BURN_UP_STEP on when it follows the student by EMA (EMA_KEEP_RATE_OFFLINE != 1)."""
        self.iter -= 1  # hooks see the index of the step just finished
        try:
            cfg, period = self.cfg, self.cfg.TEST.EVAL_PERIOD
            if self.ap_50_student is None:
                self.ap_50_student, self.ap_50_offline_teacher = {}, {}

            def eval_teacher():
                self._last_eval_results_teacher = self._evaluate(self.offline_teacher)
                self.ap_50_offline_teacher[self.iter] = self._last_eval_results_teacher["bbox"]["AP50"]

            if self._eval_due():
                self._last_eval_results_student = self._evaluate(self.model)
                self.ap_50_student[self.iter] = self._last_eval_results_student["bbox"]["AP50"]
                if self.iter == period - 1:
                    eval_teacher()
                elif self.iter <= cfg.CLOUD.BURN_UP_STEP and (self.iter - period) in self.ap_50_offline_teacher:
                    self.ap_50_offline_teacher[self.iter] = self.ap_50_offline_teacher[self.iter - period]
            teacher_start = cfg.CLOUD.BURN_UP_STEP if cfg.CLOUD.EMA_KEEP_RATE_OFFLINE != 1.0 else 10 ** 10
            if self._eval_due(start=teacher_start) and self.iter not in self.ap_50_offline_teacher:
                eval_teacher()
        finally:
            self.iter += 1

    def train(self):
        for _ in range(self.start_iter, self.max_iter):
            rec = self.run_step()
            self.before_checkpoint()
            self.after_step()
            self.prepare_next()
            m = self._write_metrics(rec, self.iter)
            if m is not None and self.rank == 0:
                print(f"iter {self.iter}: " + "  ".join(f"{k} {v:.4f}" for k, v in m.items()), flush=True)
