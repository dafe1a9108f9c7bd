"""OpenVocabularyRCNN (registered under the reference's name) and the model builders.

Mirrors coin/modeling/meta_arch/clip_rcnn.py:187-426 and coin/modeling/meta_arch/build.py:7-78
(``build_model`` dispatch on ``cfg.CLOUD.Trainer``).  ``preprocess_image`` replaces the reference's per-image
GPU -> CPU -> numpy -> GPU bounce (clip_rcnn.py:295-296) with one HIP launch per image that normalises and
zero-pads the uint8 pixels straight into the channels-last batch (``coin_normalize_pad``).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import kernels as K
from .. import layers as L
from .. import streams as _streams
from .._lib import COIN_NHWC
from ..box_ops import detector_postprocess
from ..registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY
from ..structures import Boxes, ImageList, Instances
from .roi_heads import build_roi_heads
from .rpn import build_proposal_generator


def build_backbone(cfg):
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, None)


@META_ARCH_REGISTRY.register()
class OpenVocabularyRCNN(nn.Module):
    def __init__(self, *, backbone, proposal_generator, roi_heads, pixel_mean, pixel_std, device="cuda", vis_period=0,
                 input_format="RGB", compute_dtype=torch.bfloat16):
        super().__init__()
        self.backbone, self.proposal_generator, self.roi_heads = backbone, proposal_generator, roi_heads
        self.target_device = device
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std), False)
        self._mean, self._std = [float(v) for v in pixel_mean], [float(v) for v in pixel_std]
        self.input_format, self.vis_period = input_format, vis_period
        self.set_compute_dtype(compute_dtype)

    def set_sync_free(self, flag: bool):
        """Sync-free pre_train step (random-key sampling, fixed shapes): see DualTeacherRPN.sync_free."""
        if self.proposal_generator is not None:
            self.proposal_generator.sync_free = bool(flag)

    def set_sync_free_step(self, flag: bool):
        """Sync-free step_one / step_two (fixed-shape samplers, packed losses): see DualTeacherRPN.sync_free_step."""
        if self.proposal_generator is not None:
            self.proposal_generator.sync_free_step = bool(flag)

    def set_compute_dtype(self, dtype: torch.dtype):
        assert dtype in (torch.bfloat16, torch.float32)
        self.compute_dtype = dtype
        self.roi_heads.compute_dtype = dtype

    @classmethod
    def from_config(cls, cfg):
        import os

        L.CONV_GEMM["enabled"] = bool(cfg.AMD.CONV_GEMM) and os.environ.get("COIN_CONV_GEMM", "1") != "0"
        L.CONV_GEMM["wgrad"] = (bool(cfg.AMD.CONV_GEMM_WGRAD) or os.environ.get("COIN_CONV_WGRAD") == "1") and os.environ.get("COIN_CONV_WGRAD") != "0"
        if os.environ.get("COIN_CONV_GEMM_MIN_ROWS"):   # measurements only
            L.CONV_GEMM["min_rows"] = int(os.environ["COIN_CONV_GEMM_MIN_ROWS"])
        backbone = build_backbone(cfg)
        if cfg.MODEL.ROI_HEADS.POOLING_TYPE != "attnpool":
            backbone.del_attnpool()
        roi_heads = build_roi_heads(cfg, backbone.output_shape(), backgroud=True, name=cfg.MODEL.ROI_HEADS.NAME)
        return cls(backbone=backbone, roi_heads=roi_heads, pixel_mean=cfg.INPUT.TEACHER_OFFLINE.PIXEL_MEAN,
                   pixel_std=cfg.INPUT.TEACHER_OFFLINE.PIXEL_STD, device=cfg.MODEL.DEVICE, input_format=cfg.INPUT.FORMAT,
                   vis_period=cfg.VIS_PERIOD, proposal_generator=build_proposal_generator(cfg, backbone.output_shape()),
                   compute_dtype=torch.bfloat16 if cfg.AMD.COMPUTE_DTYPE == "bf16" else torch.float32)._with_sync_free(
                       cfg.AMD.SYNC_FREE, cfg.AMD.SYNC_FREE_STEP)._with_step_graphs(bool(getattr(cfg.AMD, "STEP_GRAPHS", True)))

    def _with_step_graphs(self, flag: bool):
        self.step_graphs = bool(flag)
        self.roi_heads.step_graphs = bool(flag)
        return self

    def _with_sync_free(self, flag, flag_step=False):
        self.set_sync_free(flag)
        self.set_sync_free_step(flag_step)
        return self

    @property
    def device(self):
        return self.target_device

    def to(self, device, *args, **kwargs):
        result = super().to(device, *args, **kwargs)
        self.target_device = device
        return result

    overlap_streams = True

    # ---- look-ahead of the frozen stages -------------------------------------------------------------------------------
    # The stem and the frozen stages (FREEZE_AT) see only the images: their output for the NEXT batch does not depend on
    # this step's weight update.  A trainer announces the next batch with `set_lookahead`; this step's forward then issues
    # that work on the side stream right after its own backbone, so it executes while the main stream is in the proposal
    # chain (top-k, NMS scan, RoI sampling: a few workgroups on a 256-CU chip).  The next forward picks the result up.
    _lookahead_batch = None
    _lookahead_ready = None

    def set_lookahead(self, batched_inputs):
        self._lookahead_batch = batched_inputs

    @staticmethod
    def _batch_key(batched_inputs):
        return tuple(x["image"].data_ptr() for x in batched_inputs)

    def _launch_lookahead(self):
        nxt, self._lookahead_batch = self._lookahead_batch, None
        if nxt is None or not self.overlap_streams or not self.pixel_mean.is_cuda or not hasattr(self.backbone, "frozen_forward"):
            return
        if getattr(self, "_look_stream", None) is None:  # its own stream: the main stream later waits on `_side_stream`
            self._look_stream = _streams.role_stream(self.pixel_mean.device, "look")
        side, cur = self._look_stream, torch.cuda.current_stream()
        side.wait_stream(cur)  # starts once this step's backbone has run
        with torch.cuda.stream(side), torch.no_grad():
            imgs = self.preprocess_image(nxt)
            with self._autocast():
                x = self.backbone.frozen_forward(imgs.tensor)
        self._lookahead_ready = (self._batch_key(nxt), imgs, x, side)

    def _take_lookahead(self, batched_inputs):
        ready, self._lookahead_ready = self._lookahead_ready, None
        if ready is None or ready[0] != self._batch_key(batched_inputs):
            return None, None
        _, imgs, x, side = ready
        cur = torch.cuda.current_stream()
        cur.wait_stream(side)
        x.record_stream(cur)
        imgs.tensor.record_stream(cur)
        return imgs, x

    def _overlap_side_work(self, images, rpn_targets):
        """Work of the pre_train step that does not depend on the network's activations -- the prompt-conditioned text encoder
        (~300 small launches) and the anchor labelling / sampling (~150) -- is issued on a second HIP stream so that it runs
        concurrently with the backbone convolutions instead of serialising ~3 ms of tiny kernels on the main stream.
        Autograd replays the text encoder's backward on the same side stream.  Returns the stream to wait on (or None).
        (Round 6: issuing the text encoder AFTER the backbone, so that its launches fall into the proposal chain's sort / NMS window where
        a few workgroups are busy, and waiting for the embeddings where the classifier consumes them, was measured: 30.58 vs 30.51 ms per
        step, three interleaved pairs -- no gain, not kept.)"""
        pg, bp = self.proposal_generator, self.roi_heads.box_predictor
        if not (self.overlap_streams and images.tensor.is_cuda and pg is not None and pg.sync_free and hasattr(self.backbone, "encoder")):
            return None
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = _streams.role_stream(images.tensor.device, "side")
        side, cur = self._side_stream, torch.cuda.current_stream()
        side.wait_stream(cur)
        hw = self.backbone.encoder.visual.res4_hw(*images.tensor.shape[-2:])
        with torch.cuda.stream(side):
            text = bp.prefetch_text()
            labels, boxes = pg.prefetch_labels(hw, images.tensor.device, rpn_targets)
        for t in [text, *labels, *boxes]:
            t.record_stream(cur)
        return side

    def _autocast(self):
        return torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.compute_dtype == torch.bfloat16)

    # ---- the trainable stages of the backbone as one pair of HIP graphs (coin_amd/graphs.py) --------------------------------------
    step_graphs = True
    _seg_backbone = None

    def _backbone_features(self, images, frozen_out):
        """backbone(images) of the training forward.  C4 CLIP backbone on the GPU: the stages without trainable parameters run as before
        (or arrive from the look-ahead); the trainable stages go through a `GraphedSegment` (eager for the first calls of a shape, then
        one graph launch forward and one backward) -- as far as every convolution of them runs on the hand-written kernels: the library's
        convolutions must not be recorded into a graph on this stack (coin_amd.layers._no_library_conv_under_capture), so a stage with
        a width those kernels do not serve, and every stage before it, stays eager."""
        bb = self.backbone
        vis = getattr(getattr(bb, "encoder", None), "visual", None)
        if not (self.step_graphs and images.tensor.is_cuda and type(bb).__name__ == "CLIP_IMAGE" and hasattr(bb, "frozen_forward")
                and list(vis._out_features) == ["res4"] and vis.freeze_at <= 3 and self.compute_dtype == torch.bfloat16):
            return bb(frozen_out, frozen_done=True) if frozen_out is not None else bb(images.tensor)
        if self._seg_backbone is None:
            from ..graphs import GraphedSegment

            first = 4 + 1   # stage numbers as in ModifiedResNet.forward_stages: 3 = layer2, 4 = layer3
            for no in (4, 3):
                stage = (vis.layer2, vis.layer3)[no - 3]
                if no <= vis.freeze_at or not L.library_free([m for m in stage.modules() if isinstance(m, torch.nn.Conv2d)]):
                    break
                first = no
            if first > 4:
                self._seg_backbone = False
            else:
                mods = [(vis.layer2, vis.layer3)[no - 3] for no in range(first, 5)]

                def stretch(x):
                    with L.conv_gemm_everywhere():
                        for m in mods:
                            x = m(x)
                        return x

                self._seg_backbone = (first, GraphedSegment("backbone", stretch, lambda: [p for m in mods for p in m.parameters()],
                                                            lambda: [b for m in mods for b in m.buffers()]))
        if self._seg_backbone is False:
            return bb(frozen_out, frozen_done=True) if frozen_out is not None else bb(images.tensor)
        if frozen_out is None:
            frozen_out = bb.frozen_forward(images.tensor)
        first, seg = self._seg_backbone
        x = vis.forward_stages(frozen_out, first=vis.freeze_at + 1, last=first - 1)     # trainable stages in front of the captured ones: eager
        return {"res4": seg(x, key_extra=(bb.training, vis.layer2.training, vis.layer3.training, vis.freeze_at))}

    def preprocess_image(self, batched_inputs: List[Dict]) -> ImageList:
        imgs = [x["image"].to(self.pixel_mean.device, non_blocking=True).contiguous() for x in batched_inputs]
        batch, sizes = K.normalize_pad(imgs, self._mean, self._std, self.backbone.size_divisibility, COIN_NHWC, self.compute_dtype)
        return ImageList(batch.permute(0, 3, 1, 2), sizes)  # logical NCHW, channels-last bytes

    def forward(self, batched_inputs, merge_module=None, dual_teacher_instances=None, branch=None, step_two_data=None,
                update_prototype=False):
        if not self.training or branch == "test":
            return self.inference(batched_inputs, branch=branch)
        dev = self.pixel_mean.device
        images, frozen_out = self._take_lookahead(batched_inputs)
        if images is None:
            images = self.preprocess_image(batched_inputs)
        with self._autocast():
            if branch == "pre_train":
                rcnn = [x["RCNN"].to(dev) for x in batched_inputs]
                rpn = [x["RPN"].to(dev) for x in batched_inputs]
                merge_module = None
                side = self._overlap_side_work(images, rpn)
            features = self._backbone_features(images, frozen_out)
            self._launch_lookahead()
            if branch == "pre_train":
                if side is not None:
                    torch.cuda.current_stream().wait_stream(side)
            elif branch in ("step_one", "step_two"):
                assert dual_teacher_instances is not None, "dual_teacher_instances must not be None when brach is step_one and step_two"
                rcnn, rpn = dual_teacher_instances
            else:
                raise NotImplementedError
            if self.proposal_generator is not None:
                proposals, proposal_losses = self.proposal_generator(images, features, rpn, branch=branch)
            else:
                proposals, proposal_losses = [x["proposals"].to(dev) for x in batched_inputs], {}
            with self.roi_heads.box_predictor.shared_text():  # step branches classify twice per forward: one text-encoder pass
                _, detector_losses = self.roi_heads(images, features, proposals, self.backbone.layer4, self.backbone.attnpool, branch=branch,
                                                    merge_module=merge_module, targets=rcnn, update_prototype=update_prototype)
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses

    # ------------------------------------------------------------------ inference in two halves (the EMA teacher's pass)
    # `inference_begin` enqueues everything up to the per-image score filter with FIXED shapes and without a host round trip
    # (packed RPN proposals: exactly POST_NMS_TOPK_TEST rows per image + a validity mask; rows without a proposal get probability
    # 0 and never pass the score threshold); `inference` picks the begun pass up and runs the reference's per-image filter /
    # class-wise NMS / top-k on it (fast_rcnn.py:116-175).  CoinTrainer issues the first half for batch i+1 BEFORE the student's
    # step i whenever the teacher is not due for an EMA, so that the teacher's device work and read-back hide under that step.
    _begun = None

    def _static_inference_ok(self) -> bool:
        return self.proposal_generator is not None and type(self.roi_heads).__name__ == "OpenVocabularyRes5ROIHeads" and self.pixel_mean.is_cuda

    def _inference_core(self, x, image_sizes, branch="test"):
        """Device-only, fixed-shape part of the inference pass: x = the normalised / padded batch (logical NCHW, channels-last bytes)
        -> (boxes [N, P, 4k], probabilities [N, P, K+1] with zero rows where the RPN kept fewer than P proposals)."""
        images = ImageList(x, image_sizes)
        with self._autocast():
            features = self.backbone(x)
            packed, _ = self.proposal_generator(images, features, None, branch, packed=True)
            n, p = packed.boxes.shape[:2]
            bidx = torch.arange(n, device=packed.boxes.device, dtype=packed.boxes.dtype).repeat_interleave(p).unsqueeze(1)
            rois = torch.cat([bidx, packed.boxes.reshape(-1, 4)], dim=1)
            feats = self.roi_heads._pooled(features, rois, self.backbone.layer4, self.backbone.attnpool, fixed_shape=True)
            scores, deltas = self.roi_heads.box_predictor(feats, branch=branch)
        bp = self.roi_heads.box_predictor
        boxes = bp.box2box_transform.apply_deltas(deltas.float(), rois[:, 1:])
        probs = F.softmax(scores.float(), dim=-1) * packed.valid.reshape(-1, 1).to(torch.float32)
        return boxes.view(n, p, -1), probs.view(n, p, -1)

    # The fixed-shape half as ONE HIP graph (cfg.AMD.TEACHER_GRAPH; CoinTrainer uses it for the EMA-due iterations, whose teacher
    # pass follows the optimizer step on the device anyway and can therefore be replayed from the default stream): ~900 launches
    # of the frozen teacher -> one graph launch.  The graph re-reads the weights in place (the EMA kernel writes through the same
    # storage); the refresh of the bf16 weight shadows and the prompt transformer are captured INSIDE the graph (both are marked stale
    # before the capture), so a replay after an EMA sees the new weights without a host-side walk over the parameters.
    _graphs = None
    _graph_seen = None
    graph_failed = False

    @torch.no_grad()
    def inference_begin(self, batched_inputs, branch="test", graph: bool = False) -> bool:
        assert (not self.training) or branch == "test"
        if not self._static_inference_ok():
            return False
        images = self.preprocess_image(batched_inputs)
        sizes = tuple(tuple(int(v) for v in s) for s in images.image_sizes)
        if graph and not self.graph_failed and not torch.cuda.is_current_stream_capturing():
            out = self._graph_replay(images.tensor, sizes, branch)
            if out is not None:
                self._begun = (batched_inputs, out[0], out[1], images.image_sizes)
                return True
        boxes, probs = self._inference_core(images.tensor, images.image_sizes, branch)
        self._begun = (batched_inputs, boxes, probs, images.image_sizes)
        return True

    def _graph_replay(self, x, sizes, branch):
        key = (tuple(x.shape), tuple(x.stride()), x.dtype, sizes, branch)
        if self._graphs is None:
            self._graphs, self._graph_seen = {}, {}
        ent = self._graphs.get(key)
        if ent is None:
            n = self._graph_seen.get(key, 0) + 1
            self._graph_seen[key] = n
            if n < 3 or len(self._graphs) >= 4:   # capture only shapes that repeat (real data: a few resized sizes), at most four
                return None
            try:
                ent = self._graph_capture(x, sizes, branch)
            except Exception as e:  # same kernels either way: eager launches
                import warnings

                warnings.warn(f"teacher inference: HIP graph capture failed ({type(e).__name__}: {e}); running it eagerly")
                self.graph_failed = True
                # the aborted capture RECORDED the shadow refreshes and the text encoding without running them, yet marked them
                # fresh (and the cached text features live in the dead graph's pool): stale again before the eager pass reads them
                self._invalidate_derived()
                return None
            self._graphs[key] = ent
        static_in, g, boxes, probs = ent[:4]
        static_in.copy_(x)
        g.replay()
        return boxes, probs

    def _invalidate_derived(self):
        """bf16 weight shadows, frozen-norm constants and cached text features of this model are stale."""
        L.invalidate_storage({t.data_ptr() for t in list(self.parameters()) + list(self.buffers())})
        L.invalidate_shadows(list(self.parameters()))
        for m in self.modules():
            if hasattr(m, "invalidate_text_cache"):
                m.invalidate_text_cache()

    def _graph_capture(self, x, sizes, branch):
        static_in = x.clone()
        for _ in range(2):   # library solver searches, workspaces, anchor / size caches: everything that allocates or syncs happens here
            self._inference_core(static_in, sizes, branch)
        torch.cuda.synchronize()
        self._invalidate_derived()   # -> shadow refreshes + prompt transformer are captured: every replay re-derives them
        g = torch.cuda.CUDAGraph()
        cap = K.capture_stream_value()
        K.take_stream_workspaces(cap)    # the graph owns the workspaces its kernels are recorded with (kernels.take_stream_workspaces)
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # loader threads may touch the device meanwhile
                boxes, probs = self._inference_core(static_in, sizes, branch)
        finally:
            ws = K.take_stream_workspaces(cap)
        return static_in, g, boxes, probs, ws

    def _inference_finish(self, begun, do_postprocess=True):
        from .fast_rcnn import fast_rcnn_inference_single_image

        batched_inputs, boxes, probs, sizes = begun
        bp = self.roi_heads.box_predictor
        results = [fast_rcnn_inference_single_image(boxes[i], probs[i], tuple(sizes[i]), bp.test_score_thresh, bp.test_nms_thresh, bp.test_topk_per_image)[0]
                   for i in range(len(batched_inputs))]
        if not do_postprocess:
            return results
        return [{"instances": detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))}
                for r, inp, size in zip(results, batched_inputs, sizes)]

    @torch.no_grad()
    def inference(self, batched_inputs, branch=None, detected_instances=None, do_postprocess=True):
        assert (not self.training) or branch == "test"
        begun, self._begun = self._begun, None
        if begun is not None and begun[0] is batched_inputs:
            return self._inference_finish(begun, do_postprocess)
        images = self.preprocess_image(batched_inputs)
        with self._autocast():
            features = self.backbone(images.tensor)
            if self.proposal_generator is not None:
                proposals, _ = self.proposal_generator(images, features, None, branch)
            else:
                proposals = [x["proposals"].to(self.pixel_mean.device) for x in batched_inputs]
            results, _ = self.roi_heads(images, features, proposals, self.backbone.layer4, self.backbone.attnpool, branch=branch, targets=None)
        if not do_postprocess:
            return results
        out = []
        for r, inp, size in zip(results, batched_inputs, images.image_sizes):
            out.append({"instances": detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out


@META_ARCH_REGISTRY.register()
class CLIP(nn.Module):
    """The CLIP teacher of the pre-training data collection (clip_rcnn.py:40-151): every box cached from the cloud detector is
    re-scored by CLIP (RoIAlign -> res5 -> attention pooling -> cosine to the class embeddings, background included) and the
    boxes CLIP calls background are dropped.  `forward(batched_inputs, pre_result)` with one image, as the reference."""

    def __init__(self, *, backbone, roi_heads, pixel_mean, pixel_std, device="cuda", compute_dtype=torch.float32):
        super().__init__()
        self.backbone, self.roi_heads, self.target_device = backbone, roi_heads, device
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std), False)
        self._mean, self._std = [float(v) for v in pixel_mean], [float(v) for v in pixel_std]
        self.compute_dtype = compute_dtype

    @classmethod
    def from_config(cls, cfg):
        from .text_encoder import text_dim_of

        backbone = build_backbone(cfg)
        if backbone.attnpool is None:
            backbone.add_attnpool(text_dim_of(cfg))
        roi_heads = build_roi_heads(cfg, backbone.output_shape(), backgroud=True, name=cfg.MODEL.ROI_HEADS.TEACHER_OFFLINE)
        return cls(backbone=backbone, roi_heads=roi_heads, pixel_mean=cfg.INPUT.TEACHER_OFFLINE.PIXEL_MEAN,
                   pixel_std=cfg.INPUT.TEACHER_OFFLINE.PIXEL_STD, device=cfg.MODEL.DEVICE,
                   compute_dtype=torch.bfloat16 if cfg.AMD.COMPUTE_DTYPE == "bf16" else torch.float32)

    @property
    def device(self):
        return self.target_device

    @torch.no_grad()
    def forward(self, batched_inputs, pre_result):
        assert len(batched_inputs) == 1
        b = batched_inputs[0]
        for k in ("file_name", "height", "width", "image_id"):
            assert pre_result[k] == b[k]
        img = b["image"].to(self.pixel_mean.device).contiguous()
        batch, _ = K.normalize_pad([img], self._mean, self._std, self.backbone.size_divisibility, COIN_NHWC, self.compute_dtype)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.compute_dtype == torch.bfloat16 and img.is_cuda):
            features = self.backbone(batch.permute(0, 3, 1, 2))
            return self.get_clip_result(features, batched_inputs, pre_result)

    def preprocess_boxes(self, boxes: Boxes, batched_inputs) -> Boxes:
        new = Boxes(boxes.tensor.clone())
        net_h, net_w = batched_inputs[0]["image"].shape[1:]
        new.scale(net_w / batched_inputs[0]["width"], net_h / batched_inputs[0]["height"])
        return new

    def get_clip_result(self, image_features, batched_inputs, pre_result):
        assert self.backbone.attnpool is not None
        dev = self.pixel_mean.device
        out = {k: v for k, v in pre_result.items() if not isinstance(v, dict)}
        for tag in ("RCNN", "RPN", "RPN_AUG"):
            if tag not in pre_result:
                continue
            src = pre_result[tag]["instances"]
            if len(src) == 0:
                out[tag] = {"instances": src}
                continue
            inst = Instances(src.image_size)
            inst.pred_boxes = Boxes(src.pred_boxes.tensor.clone().to(dev))
            prop = Instances(src.image_size)
            prop.proposal_boxes = self.preprocess_boxes(inst.pred_boxes, batched_inputs)
            probs = self.roi_heads(image_features, [prop], self.backbone.layer4, self.backbone.attnpool)
            inst.scores, inst.pred_classes = probs.max(1)
            inst.probs = probs
            out[tag] = {"instances": inst[inst.pred_classes != probs.size(1) - 1]}  # boxes CLIP calls background are dropped
        return out


def build_model(cfg):
    """coin/modeling/meta_arch/build.py:7-78 for the trainers of the hot path.  The cloud / CLIP collectors
    (GroundingDINO, GLIP) are out of scope: their cached outputs are an input of this path."""
    dev = torch.device(cfg.MODEL.DEVICE)
    arch = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)

    def make():
        m = arch.from_config(cfg).to(dev)
        if dev.type == "cuda":
            # activations are channels-last end to end; keeping the 4-D convolution weights (fp32 masters, their gradients,
            # momentum buffers and bf16 shadows) in the same memory format removes a per-call weight re-layout from every
            # convolution forward / backward.  Logical shapes and state-dict contents are unchanged.
            nn.Module.to(m, memory_format=torch.channels_last)
        return m

    if cfg.CLOUD.Trainer in ("PRETRAIN", "ORACLE", "ModelZoo_test"):
        return make()
    if cfg.CLOUD.Trainer == "CoinTrainer":
        return make(), make()
    raise NotImplementedError(f"CLOUD.Trainer={cfg.CLOUD.Trainer!r} is outside the adaptation-training hot path")
