from .base import BASE_Trainer  # noqa: F401
from .pre_train import PRETrainer  # noqa: F401
from .trainer import CoinTrainer  # noqa: F401
