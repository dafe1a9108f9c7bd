"""OpenVocabularyRes5ROIHeads (registered under the reference's name) on the HIP RoIAlign.

Mirrors coin/modeling/roi_heads/clip_roi_heads.py:90-399: proposal labelling / sampling into (fg, bg) or
(A, B, bg), RoIAlign 14x14 on res4, ``backbone.layer4`` on the RoI tiles, mean pool, box predictor, and the
extra pass over the raw private (C) boxes.  RoIAlign is ``coin_roi_align_fwd/bwd`` on channels-last
activations; its output feeds the res5 convolutions without a layout change.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from .. import layers as L
from ..box_ops import Matcher, PackedProposals, add_ground_truth_to_proposals, sample_masks, subsample_labels
from ..registry import ROI_HEADS_REGISTRY
from ..structures import Boxes, Instances, ShapeSpec, pairwise_iou
from .fast_rcnn import FastRCNNOutputLayers
from .text_encoder import TEXT_DIMS, build_text_encoder


class PackedSamples:
    """Sampled RoIs of the sync-free pre_train path, fixed shape [N*R]: rows are (fg..., bg..., invalid...) per image.
    gt_classes: fg class | num_classes (bg) | -1 (invalid filler when an image has fewer than R candidates)."""

    def __init__(self, boxes, gt_classes, gt_boxes, gt_probs, per_image):
        self.boxes, self.gt_classes, self.gt_boxes, self.gt_probs, self.per_image = boxes, gt_classes, gt_boxes, gt_probs, per_image


class PackedStepSamples:
    """Sampled RoIs of a sync-free step_one / step_two pass, fixed shape [N*R].  role: 0 = A (consistent), 1 = B (inconsistent),
    2 = background, -1 = filler.  gt_classes: A -> class, background -> num_classes; gt_classes_online / _offline: B rows;
    gt_boxes: matched teacher box (A, B) or the proposal itself; gt_probs_online / _offline [.., K+1]: A and B rows (0 elsewhere).
    has_b: the matcher produced at least one B target in the batch (known on the host)."""

    def __init__(self, boxes, role, gt_classes, gt_classes_online, gt_classes_offline, gt_boxes, gt_probs_online, gt_probs_offline,
                 per_image, num_images, has_b):
        self.boxes, self.role, self.gt_classes = boxes, role, gt_classes
        self.gt_classes_online, self.gt_classes_offline, self.gt_boxes = gt_classes_online, gt_classes_offline, gt_boxes
        self.gt_probs_online, self.gt_probs_offline = gt_probs_online, gt_probs_offline
        self.per_image, self.num_images, self.has_b = per_image, num_images, has_b


class ROIPooler(nn.Module):
    """Single-level detectron2 ROIPooler (ROIAlignV2 = aligned)."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type="ROIAlignV2"):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        assert len(scales) == 1 and pooler_type in ("ROIAlignV2", "ROIAlign")
        self.scale, self.sampling_ratio, self.aligned = float(scales[0]), int(sampling_ratio), pooler_type == "ROIAlignV2"

    def forward(self, x: List[torch.Tensor], box_lists) -> torch.Tensor:
        if isinstance(box_lists, torch.Tensor):  # already [R, 5] rows (batch index, box)
            return L.roi_align(x[0], box_lists, self.output_size, self.scale, self.sampling_ratio, self.aligned)
        rois = torch.cat([torch.cat([b.tensor.new_full((len(b), 1), float(i)), b.tensor], dim=1) for i, b in enumerate(box_lists)], dim=0)
        return L.roi_align(x[0], rois, self.output_size, self.scale, self.sampling_ratio, self.aligned)


@ROI_HEADS_REGISTRY.register()
class OpenVocabularyRes5ROIHeads(nn.Module):
    def __init__(self, *, in_features, pooler, box_predictor, pooling_type, num_classes, batch_size_per_image, positive_fraction,
                 proposal_matcher, proposal_append_gt=True, BG_TRAIN=True, mask_head=None):
        super().__init__()
        assert mask_head is None, "MODEL.MASK_ON is False in every COIN config"
        self.in_features, self.pooler, self.box_predictor, self.pooling_type = in_features, pooler, box_predictor, pooling_type
        self.num_classes, self.batch_size_per_image, self.positive_fraction = num_classes, batch_size_per_image, positive_fraction
        self.proposal_matcher, self.proposal_append_gt, self.BG_TRAIN = proposal_matcher, proposal_append_gt, BG_TRAIN
        self.compute_dtype = torch.float32

    @classmethod
    def from_config(cls, cfg, input_shape, backgroud):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        assert not cfg.MODEL.KEYPOINT_ON and len(in_features) == 1
        text_encoder = build_text_encoder(cfg, backgroud)
        pooling_type = cfg.MODEL.ROI_HEADS.POOLING_TYPE
        res5_ch = input_shape[in_features[0]].channels * 2
        ch = res5_ch if pooling_type != "attnpool" else TEXT_DIMS[cfg.MODEL.TEACHER_OFFLINE.TYPE]
        return cls(
            in_features=in_features,
            pooler=ROIPooler(cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION, (1.0 / input_shape[in_features[0]].stride,),
                             cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO, cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE),
            box_predictor=FastRCNNOutputLayers.from_config(cfg, text_encoder, ShapeSpec(channels=ch, height=1, width=1)),
            pooling_type=pooling_type,
            num_classes=len(text_encoder.classes) - 1 if backgroud else len(text_encoder.classes),
            batch_size_per_image=cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, positive_fraction=cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            proposal_matcher=Matcher(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, cfg.MODEL.ROI_HEADS.IOU_LABELS, allow_low_quality_matches=False),
            proposal_append_gt=cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT, BG_TRAIN=cfg.CLOUD.BG_TRAIN)._set(
                inference_rois_per_image=cfg.MODEL.RPN.POST_NMS_TOPK_TEST)

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        return self

    def _shared_roi_transform(self, features, boxes, backbone_res5):
        return backbone_res5(self.pooler(features, boxes))

    ROI_BUCKET = 64

    inference_rois_per_image = None  # MODEL.RPN.POST_NMS_TOPK_TEST: the teacher / evaluation pass is padded to exactly this many rows

    def _pooled(self, features, boxes, res5, attnpool, fixed_shape: bool = False, pad_to: int = 0):
        """RoIAlign -> res5 -> pooled features [R, C].

        Passes whose RoI count changes from call to call (the C-box pass, the step_one/two proposal pass when an image yields
        fewer candidates, inference) are padded to a multiple of ROI_BUCKET rows: MIOpen compiles / searches kernels per
        convolution shape (seconds for every new batch size), so the shapes must repeat.  The padding is exact: the filler
        rows are excluded from the train-mode BatchNorm statistics and receive zero gradient (`layers.valid_rows`), and
        their outputs are dropped here."""
        if isinstance(boxes, torch.Tensor):
            rois = boxes
        else:
            rois = torch.cat([torch.cat([b.tensor.new_full((len(b), 1), float(i)), b.tensor], dim=1) for i, b in enumerate(boxes)], dim=0)
        r = rois.shape[0]
        bucket = self.ROI_BUCKET if r <= 16 * self.ROI_BUCKET else 4 * self.ROI_BUCKET
        pad = 0 if (fixed_shape or r == 0 or not rois.is_cuda) else (-r) % bucket
        if pad_to and rois.is_cuda and 0 < r <= pad_to:
            pad = pad_to - r  # one shape for ever (the EMA teacher's proposal count changes every step)
        if pad:
            rois = torch.cat([rois, rois.new_zeros((pad, 5))], dim=0)
        with L.valid_rows(r if pad else None):
            out = self._pooled_graphed(features, rois, res5, attnpool)
        return out[:r] if pad else out

    step_graphs = True
    _trunk_segs = None

    def _pooled_graphed(self, features, rois, res5, attnpool):
        """`_pooled_rows` through a `GraphedSegment` (coin_amd/graphs.py) during training on the GPU: RoIAlign -> res5 -> mean pool is a fixed
        launch sequence for a given (feature map shape, RoI count)."""
        if not (self.step_graphs and self.training and torch.is_grad_enabled() and self.pooling_type == "meanpool" and len(self.in_features) == 1
                and rois.is_cuda and rois.shape[0] > 0 and isinstance(res5, torch.nn.Sequential) and L._VALID_ROWS[0] is None):
            # (a padded pass -- the C boxes, whose count changes every step -- stays eager: each (count, padding) pair would be its own graph)
            seg = (self._trunk_segs or {}).get(id(res5))
            if seg is not None and seg[1] is not None and torch.is_grad_enabled():
                seg[1].note_outside_use()   # res5's weights get a second gradient contribution in this step: no chunk-by-chunk hand-over
            if (rois.is_cuda and rois.shape[0] > 0 and self.compute_dtype == torch.bfloat16 and isinstance(res5, torch.nn.Sequential)
                    and self._res5_library_free(res5)):
                # ... but on the hand-written kernels whatever its row count (round 6: the 128 x 128 small-map cores serve a 64-RoI pass as
                # well as the library does): a RoI bucket the library has not met before cost a solver search of tens of milliseconds in
                # the step where it first appeared -- step_two's groups of four steps read 64 ... 148 ms (tools/td_mode_check.sh)
                with L.conv_gemm_everywhere():
                    return self._pooled_rows(features, rois, res5, attnpool)
            return self._pooled_rows(features, rois, res5, attnpool)
        if self._trunk_segs is None:
            self._trunk_segs = {}
        seg = self._trunk_segs.get(id(res5))
        if seg is None or seg[0]() is not res5:
            import weakref

            from ..graphs import GraphedSegment

            name = self.in_features[0]

            def trunk(feat, r):   # every convolution on the hand-written kernels: library launches must not be recorded into a graph
                with L.conv_gemm_everywhere():
                    return self._pooled_rows({name: feat}, r, res5, attnpool)

            convs = [m for m in res5.modules() if isinstance(m, torch.nn.Conv2d)]
            ok = self.compute_dtype == torch.bfloat16 and L.library_free(convs)
            seg = (weakref.ref(res5), GraphedSegment("roi_trunk", trunk, lambda: list(res5.parameters()), lambda: list(res5.buffers())) if ok else None)
            self._trunk_segs[id(res5)] = seg
        if seg[1] is None:   # a width the hand-written weight-gradient kernel does not serve
            return self._pooled_rows(features, rois, res5, attnpool)
        return seg[1](features[self.in_features[0]], rois, key_extra=(res5.training, self.compute_dtype))

    @staticmethod
    def _res5_library_free(res5) -> bool:
        return L.library_free([m for m in res5.modules() if isinstance(m, torch.nn.Conv2d)])

    def _pooled_rows(self, features, rois, res5, attnpool):
        if self.pooling_type == "meanpool" and isinstance(res5, torch.nn.Sequential) and len(res5) > 0 and hasattr(res5[-1], "conv3"):
            # res5 (clip_roi_heads.py:172-176) with the spatial mean (:207-208) folded into the last block's epilogue
            x = self.pooler([features[f] for f in self.in_features], rois)
            for block in list(res5)[:-1]:
                x = block(x)
            return res5[-1](x, mean_pool=True).flatten(1).to(self.compute_dtype)
        x = self._shared_roi_transform([features[f] for f in self.in_features], rois, res5)
        if self.pooling_type == "meanpool":
            return x.mean(dim=[2, 3]).to(self.compute_dtype)
        if self.pooling_type == "attnpool":
            return attnpool(x).to(self.compute_dtype)
        raise NotImplementedError

    def forward(self, images, features, proposals, res5, attnpool, branch, merge_module=None, targets=None, update_prototype=False):
        train = self.training and branch != "test"
        if train and branch == "pre_train" and isinstance(proposals, PackedProposals):
            ps = self.sample_packed(proposals, targets)
            n, r = len(proposals), ps.per_image
            bidx = torch.arange(n, device=ps.boxes.device, dtype=ps.boxes.dtype).repeat_interleave(r).unsqueeze(1)
            predictions = self.box_predictor(self._pooled(features, torch.cat([bidx, ps.boxes], dim=1), res5, attnpool, fixed_shape=True), branch=branch)
            return [], self.box_predictor.losses_packed(predictions, ps, update_prototype=update_prototype)
        if train and branch in ("step_one", "step_two") and isinstance(proposals, PackedProposals):
            ta, tb, tc = [t[0] for t in targets], [t[1] for t in targets], [t[2] for t in targets]
            ps = self.sample_packed_step(proposals, ta, tb, tc)
            n, r = len(proposals), ps.per_image
            bidx = torch.arange(n, device=ps.boxes.device, dtype=ps.boxes.dtype).repeat_interleave(r).unsqueeze(1)
            predictions = self.box_predictor(self._pooled(features, torch.cat([bidx, ps.boxes], dim=1), res5, attnpool, fixed_shape=True), branch=branch)
            cpred = None
            if sum(len(c) for c in tc) != 0:  # host integers
                cpred = self.box_predictor(self._pooled(features, [c.gt_boxes for c in tc], res5, attnpool), branch=branch, return_feats=False)
            return [], self.box_predictor.losses_packed_step(predictions, ps, cpred, tc if cpred is not None else None, merge_module, branch,
                                                             update_prototype=update_prototype)
        if train:
            assert targets
            if branch == "pre_train":
                proposals = self.label_and_sample_proposals(proposals, targets, branch=branch)
                boxes = [Boxes.cat([p[0].proposal_boxes, p[1].proposal_boxes]) for p in proposals]
            elif branch in ("step_one", "step_two"):
                ta, tb, tc = [t[0] for t in targets], [t[1] for t in targets], [t[2] for t in targets]
                proposals = self.label_and_sample_proposals(proposals, [ta, tb, tc], branch=branch)
                boxes = [Boxes.cat([p[0].proposal_boxes, p[1].proposal_boxes, p[2].proposal_boxes]) for p in proposals]
            else:
                raise NotImplementedError
        else:
            boxes = [p.proposal_boxes for p in proposals]
        pad_to = 0 if train or not self.inference_rois_per_image else self.inference_rois_per_image * len(boxes)
        predictions = self.box_predictor(self._pooled(features, boxes, res5, attnpool, pad_to=pad_to), branch=branch)
        if not train:
            pred_instances, _ = self.box_predictor.inference(predictions, proposals)
            return pred_instances, {}
        if branch != "pre_train":
            if sum(len(c) for c in tc) != 0:
                cpred = self.box_predictor(self._pooled(features, [c.gt_boxes for c in tc], res5, attnpool), branch=branch, return_feats=False)
                predictions, proposals = (predictions, cpred), (proposals, tc)
            else:
                predictions, proposals = (predictions, ((None, None), None)), (proposals, None)
        return [], self.box_predictor.losses(predictions, proposals, merge_module, branch=branch, update_prototype=update_prototype)

    def _sample_proposals(self, matched_idxs, matched_labels, gt_classes):
        if gt_classes.numel() > 0:
            gt_classes = gt_classes[matched_idxs]
            gt_classes[matched_labels == 0] = self.num_classes
            gt_classes[matched_labels == -1] = -1
        else:
            gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
        fg, bg = subsample_labels(gt_classes, self.batch_size_per_image, self.positive_fraction, self.num_classes)
        sampled = torch.cat([fg, bg], dim=0)
        return sampled, gt_classes[sampled]

    @torch.no_grad()
    def sample_packed(self, proposals: PackedProposals, targets) -> PackedSamples:
        """label_and_sample_proposals(pre_train) (clip_roi_heads.py:286-340) with fixed shapes and no host round trip:
        candidates = RPN proposals (+ validity) ++ teacher boxes, same Matcher, `sample_masks` instead of randperm."""
        k, r = self.num_classes, self.batch_size_per_image
        dev = proposals.boxes.device
        n = len(targets)
        counts = [len(t) for t in targets]
        gmax = max(counts)
        # Batched over the images (round 4: ~15 launches instead of ~20 per image): the teacher boxes / classes / probabilities as
        # zero-padded [N, gmax, ...] blocks, ONE matcher launch sequence with a candidate set per image (coin_anchor_match,
        # anchors_per_image), gathers instead of per-image indexing.  Row for row the values of the per-image form.
        def padded(get, tail, dtype):
            rows = [get(t) for t in targets]
            if all(c == gmax for c in counts):
                return torch.stack(rows) if gmax else torch.zeros((n, 0) + tail, dtype=dtype, device=dev)
            out = torch.zeros((n, gmax) + tail, dtype=dtype, device=dev)
            for i, (row, c) in enumerate(zip(rows, counts)):
                if c:
                    out[i, :c] = row
            return out

        gt_b = padded(lambda t: t.gt_boxes.tensor, (4,), proposals.boxes.dtype)
        pb, pv = proposals.boxes, proposals.valid
        if self.proposal_append_gt and gmax:
            if all(c == gmax for c in counts):
                gv = torch.ones((n, gmax), dtype=torch.bool, device=dev)
            else:
                gv = torch.zeros((n, gmax), dtype=torch.bool, device=dev)
                for i, c in enumerate(counts):
                    if c:
                        gv[i, :c] = True
            pb = torch.cat([pb, gt_b], dim=1)       # padding rows are zero boxes, marked invalid
            pv = torch.cat([pv, gv], dim=1)
        pb = pb.contiguous()
        m = pb.shape[1]
        if gmax:
            # IoU + arg-max + threshold band in one launch sequence for the batch (bit-identical to Matcher(pairwise_iou(...)) per image);
            # an image without teacher boxes gets index 0 and the background band
            idx, lab, _ = self.proposal_matcher.match_boxes([t.gt_boxes.tensor for t in targets], pb, want_boxes=False)
            fgm = lab == 1
            gt_c = padded(lambda t: t.gt_classes_offline, (), torch.int64)
            gt_p = padded(lambda t: t.gt_probs_offline, (k + 1,), gt_b.dtype if gt_b.is_floating_point() else torch.float32)
            cls = torch.where(fgm, gt_c.gather(1, idx), torch.full_like(idx, k))
            gtb = torch.where(fgm.unsqueeze(-1), gt_b.gather(1, idx.unsqueeze(-1).expand(-1, -1, 4)), pb)
            prs = gt_p.gather(1, idx.unsqueeze(-1).expand(-1, -1, k + 1))
        else:
            cls = torch.full((n, m), k, dtype=torch.int64, device=dev)
            gtb, prs = pb, pb.new_zeros(n, m, k + 1)
        cls = torch.where(pv, cls, torch.full_like(cls, -1))             # [N, M]
        boxes = pb
        fg, bg = sample_masks(cls, r, self.positive_fraction, k)
        # exactly r rows per image: chosen fg first, then chosen bg, then (only if short) unchosen fillers marked invalid
        prio = torch.where(fg, 0, torch.where(bg, 1, 2))
        sel = prio.argsort(dim=1, stable=True)[:, :r]
        chosen = (fg | bg).gather(1, sel)
        out_cls = torch.where(chosen, cls.gather(1, sel), torch.full_like(sel, -1))
        g4 = sel.unsqueeze(-1).expand(-1, -1, 4)
        return PackedSamples(boxes.gather(1, g4).reshape(-1, 4), out_cls.reshape(-1), gtb.gather(1, g4).reshape(-1, 4),
                             prs.gather(1, sel.unsqueeze(-1).expand(-1, -1, k + 1)).reshape(-1, k + 1), min(r, sel.shape[1]))

    @torch.no_grad()
    def sample_packed_step(self, proposals: PackedProposals, ta, tb, tc) -> PackedStepSamples:
        """label_and_sample_proposals(step_one / step_two) (clip_roi_heads.py:341-399) with fixed shapes and no host round trip:
        candidates = RPN proposals (+ validity) ++ A boxes ++ B boxes, the same Matcher against cat(A, B, C), candidates that match a
        private (C) box with IoU >= thr are ignored, `sample_masks` instead of randperm.  The target counts are host integers (the
        matcher ran on the host), only the candidates' roles are data dependent and they stay on the device."""
        k, r = self.num_classes, self.batch_size_per_image
        dev = proposals.boxes.device
        extra = max(len(a) + len(b) for a, b in zip(ta, tb)) if self.proposal_append_gt else 0
        rows = {n: [] for n in ("box", "cls", "role", "con", "coff", "gtb", "pon", "poff")}
        for i, (a, b, c) in enumerate(zip(ta, tb, tc)):
            la, lb, lc = len(a), len(b), len(c)
            pb, pv = proposals.boxes[i], proposals.valid[i]
            if self.proposal_append_gt:
                pad = extra - la - lb
                pb = torch.cat([pb, a.gt_boxes.tensor, b.gt_boxes.tensor, pb.new_zeros(pad, 4)])
                pv = torch.cat([pv, torch.ones(la + lb, dtype=torch.bool, device=dev), torch.zeros(pad, dtype=torch.bool, device=dev)])
            m = pb.shape[0]
            if la + lb + lc > 0:
                tboxes = Boxes.cat([a.gt_boxes, b.gt_boxes, c.gt_boxes])
                idx, lab, _ = self.proposal_matcher.match_boxes([tboxes.tensor], pb, want_boxes=False)   # one fused launch
                idx, lab = idx[0], lab[0]
                fg = lab == 1
                in_a, in_b = idx < la, (idx >= la) & (idx < la + lb)
                role = torch.where(fg & in_a, 0, torch.where(fg & in_b, 1, torch.where(fg, -1, 2)))  # fg on a C box -> ignored
                z = lambda n, *shape: pb.new_zeros((n,) + shape)
                zl = lambda n: torch.zeros(n, dtype=torch.int64, device=dev)
                con_t = torch.cat([a.gt_classes, b.gt_classes_online, zl(lc)])[idx]
                coff_t = torch.cat([a.gt_classes, b.gt_classes_offline, zl(lc)])[idx]
                pon_t = torch.cat([a.gt_probs_online, b.gt_probs_online, z(lc, k + 1)])[idx]
                poff_t = torch.cat([a.gt_probs_offline, b.gt_probs_offline, z(lc, k + 1)])[idx]
                gtb = torch.where((role == 0).unsqueeze(1) | (role == 1).unsqueeze(1), tboxes.tensor[idx], pb)
            else:
                role = torch.full((m,), 2, dtype=torch.int64, device=dev)
                con_t = coff_t = torch.zeros(m, dtype=torch.int64, device=dev)
                pon_t = poff_t = pb.new_zeros(m, k + 1)
                gtb = pb
            role = torch.where(pv, role, torch.full_like(role, -1))
            cls = torch.where(role == 0, con_t, torch.where(role == 1, con_t, torch.where(role == 2, torch.full_like(role, k), torch.full_like(role, -1))))
            for n, v in (("box", pb), ("cls", cls), ("role", role), ("con", con_t), ("coff", coff_t), ("gtb", gtb), ("pon", pon_t), ("poff", poff_t)):
                rows[n].append(v)
        t = {n: torch.stack(v) for n, v in rows.items()}                     # [N, M, ...]
        fg, bg = sample_masks(t["cls"], r, self.positive_fraction, k)          # positives = A or B rows (online class), negatives = bg
        if not self.BG_TRAIN:
            bg = torch.zeros_like(bg)
        prio = torch.where(fg, 0, torch.where(bg, 1, 2))
        sel = prio.argsort(dim=1, stable=True)[:, :r]
        chosen = (fg | bg).gather(1, sel)
        g1 = lambda x: x.gather(1, sel).reshape(-1)
        gn = lambda x: x.gather(1, sel.unsqueeze(-1).expand(-1, -1, x.shape[-1])).reshape(-1, x.shape[-1])
        role = torch.where(chosen, t["role"].gather(1, sel), torch.full_like(sel, -1)).reshape(-1)
        gt_cls = torch.where(role == 2, torch.full_like(role, k), g1(t["con"]))   # A rows: their class; bg rows: num_classes
        return PackedStepSamples(gn(t["box"]), role, gt_cls, g1(t["con"]), g1(t["coff"]), gn(t["gtb"]), gn(t["pon"]), gn(t["poff"]),
                                 min(r, sel.shape[1]), len(ta), any(len(b) > 0 for b in tb))

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets, branch):
        """clip_roi_heads.py:282-399 (the `no_thresh_boxes` variant is never produced by the trainers, base.py:119-121)."""
        out = []
        if branch == "pre_train":
            if self.proposal_append_gt:
                proposals = add_ground_truth_to_proposals(targets, proposals)
            for p, t in zip(proposals, targets):
                idx, lab = self.proposal_matcher(pairwise_iou(t.gt_boxes, p.proposal_boxes))
                sampled, cls = self._sample_proposals(idx, lab, t.gt_classes_offline)
                idx = idx[sampled]
                is_bg = cls == self.num_classes
                fg, bg = p[sampled[~is_bg]], p[sampled[is_bg]]
                bg.gt_classes = cls[is_bg]
                for name, val in t.get_fields().items():
                    if name.startswith("gt_") and not fg.has(name):
                        fg.set(name, val[idx[~is_bg]])
                out.append((fg, bg))
            return out
        ta, tb, tc = targets
        if self.proposal_append_gt:
            proposals = add_ground_truth_to_proposals(ta, proposals)
            proposals = add_ground_truth_to_proposals(tb, proposals)
        for p, a, b, c in zip(proposals, ta, tb, tc):
            la, lb, lc = len(a), len(b), len(c)
            idx, lab = self.proposal_matcher(pairwise_iou(Boxes.cat([a.gt_boxes, b.gt_boxes, c.gt_boxes]), p.proposal_boxes))
            in_c = (idx >= la + lb) & (idx < la + lb + lc)
            lab[in_c & (lab != 0)] = -1
            sampled, cls = self._sample_proposals(idx, lab, torch.cat([a.gt_classes, b.gt_classes_online, c.gt_classes]))
            idx = idx[sampled]
            is_bg = cls == self.num_classes
            m_a = (idx >= 0) & (idx < la) & ~is_bg
            m_b = (idx >= la) & (idx < la + lb) & ~is_bg
            pa, pb, pg = p[sampled[m_a]], p[sampled[m_b]], p[sampled[is_bg]]
            pg.gt_classes = cls[is_bg]
            if not self.BG_TRAIN:
                pg = pg[0:0]
            for name, val in a.get_fields().items():
                if name.startswith("gt_") and not pa.has(name):
                    pa.set(name, val[idx[m_a]])
            for name, val in b.get_fields().items():
                if name.startswith("gt_") and not pb.has(name):
                    pb.set(name, val[idx[m_b] - la])
            out.append((pa, pb, pg))
        return out


@ROI_HEADS_REGISTRY.register()
class CLIPRes5ROIHeads(nn.Module):
    """The CLIP teacher's head (clip_roi_heads.py:19-87): boxes from outside -> RoIAlign -> res5 -> attention (or mean) pooling ->
    softmax(exp(logit_scale) * cosine) against the fixed class embeddings.  Inference only."""

    def __init__(self, *, in_features, pooler, text_encoder):
        super().__init__()
        self.in_features, self.pooler, self.text_encoder = in_features, pooler, text_encoder

    @classmethod
    def from_config(cls, cfg, input_shape, backgroud):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        assert not cfg.MODEL.KEYPOINT_ON and len(in_features) == 1
        return cls(in_features=in_features,
                   pooler=ROIPooler(cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION, (1.0 / input_shape[in_features[0]].stride,),
                                    cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO, cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE),
                   text_encoder=build_text_encoder(cfg, backgroud))

    def forward(self, features, proposals, res5=None, attnpool=None):
        x = res5(self.pooler([features[f] for f in self.in_features], [p.proposal_boxes for p in proposals]))
        region = attnpool(x) if attnpool is not None else x.mean(dim=[2, 3])
        return self.do_classify(region.float())

    def do_classify(self, image_features):
        text = self.text_encoder(added=False).float()
        image_features = image_features / image_features.norm(dim=1, keepdim=True)
        text = text / text.norm(dim=1, keepdim=True)
        return (self.text_encoder.logit_scale.exp() * image_features @ text.t()).softmax(dim=-1)


def build_roi_heads(cfg, input_shape, backgroud=False, name=None):
    return ROI_HEADS_REGISTRY.get(name or cfg.MODEL.ROI_HEADS.NAME).from_config(cfg, input_shape, backgroud)
