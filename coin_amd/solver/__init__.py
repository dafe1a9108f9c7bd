from .build import FusedSGD, build_lr_scheduler, build_optimizer, get_default_optimizer_params  # noqa: F401
from .lr_scheduler import WarmupTwoStageMultiStepLR  # noqa: F401
