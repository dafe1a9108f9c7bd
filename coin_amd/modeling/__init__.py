from .backbone import CLIP_IMAGE, build_clip_image_backbone  # noqa: F401
from .fast_rcnn import FastRCNNOutputLayers  # noqa: F401
from .meta_arch import OpenVocabularyRCNN, build_backbone, build_model  # noqa: F401
from .roi_heads import OpenVocabularyRes5ROIHeads, build_roi_heads  # noqa: F401
from .rpn import DualTeacherRPN, build_proposal_generator  # noqa: F401
from .text_encoder import CKGNet, CLIP_TEXT, build_merge, build_text_encoder  # noqa: F401
from .fpn import CLIPResNetFPN, OpenVocabularyFPNROIHeads, SwinFPN, build_clip_resnet_fpn_backbone, build_swint_fpn_backbone  # noqa: F401,E402
