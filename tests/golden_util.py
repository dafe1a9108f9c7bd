"""Helpers shared by the golden-vector tests: rebuild oracle objects from the stored arrays."""
import os

import numpy as np
import torch

from oracle import coin as OC
from oracle import d2

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = 3
TEXT_DIM, CTX = 32, 16


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def T(a):
    t = torch.from_numpy(np.asarray(a))
    return t


def instances(z, prefix, size):
    inst = d2.Instances(tuple(size))
    for k in z.files:
        if k.startswith(prefix + "."):
            name = k[len(prefix) + 1:]
            if "." in name:
                continue
            v = T(z[k])
            inst.set(name, d2.Boxes(v) if name.endswith("boxes") else v)
    return inst


def tiny_tokens():
    toks = torch.zeros(K + 1, CTX, dtype=torch.int)
    for i in range(K + 1):
        seq = [62, 1, 2, 3, 1, 6, 6, 6, 6, 10 + i, 5, 63]
        toks[i, : len(seq)] = torch.tensor(seq)
    return toks


def tiny_text_encoder():
    enc = OC.TextEncoder(TEXT_DIM, CTX, 64, 32, 2, 2, tiny_tokens(), 4, 4)
    return OC.ClipText(enc, ["car", "person", "bus", "backgroud"], torch.randn(K + 1, TEXT_DIM))


LOSS_W = {"loss_box_reg": 1.0, "loss_box_reg_offline": 1.0, "loss_box_reg_online": 1.0, "loss_cls": 1.0,
          "loss_text_align": 10.0, "loss_distillation": 0.1, "loss_cls_b": 0.1}


def tiny_box_predictor(in_ch, dataset=("foggytrain_0.02",), loss_type="MILCrossEntropy"):
    return OC.BoxPredictor(in_ch, tiny_text_encoder(), TEXT_DIM, [1.0] * K + [0.9], LOSS_W, 32, cls_b_thresh=0.3, dataset=dataset,
                           loss_type=loss_type)


def tiny_detector():
    """Same architecture as tests/golden/gen_golden.py::build_detector, built from the oracle classes."""
    bb = OC.ClipImageBackbone(layers=(1, 1, 2, 2), width=8, freeze_at=2, update_backbone=True, zero_init_bn3=False)
    bp = tiny_box_predictor(256)
    rh = OC.Res5ROIHeads(bp, K, batch_size_per_image=32, positive_fraction=0.25)
    pg = OC.DualTeacherRPN(128, anchor_sizes=((32, 64, 128),), batch_size_per_image=64, pre_nms_topk=(200, 120), post_nms_topk=(60, 40))
    return OC.OpenVocabularyRCNN(bb, pg, rh)


def load_weights(module, z, prefix="w::", strict=True):
    sd = {k[len(prefix):]: T(z[k]) for k in z.files if k.startswith(prefix)}
    own = module.state_dict()
    missing = [k for k in own if k not in sd]
    extra = [k for k in sd if k not in own]
    if strict:
        assert not missing, f"missing keys {missing[:5]}"
        assert not extra, f"unexpected keys {extra[:5]}"
    module.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return module


def close(a, b, tol=1e-4, what=""):
    a = torch.as_tensor(np.asarray(a)).double() if not isinstance(a, torch.Tensor) else a.detach().cpu().double()
    b = torch.as_tensor(np.asarray(b)).double() if not isinstance(b, torch.Tensor) else b.detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:.0e} * {scale:.3g}"
