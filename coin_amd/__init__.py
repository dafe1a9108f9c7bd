"""coin_amd: MI355X-native implementation of COIN's adaptation-training hot path.

Only what the path needs: ``csrc/`` (HIP kernels + C ABI, ``include/coin_hip.h``), ``kernels``
(tensor-level binding) and the host-side mirror of the reference's registries / modules.
"""
__version__ = "0.1.0"

import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  This package runs 3-4 streams of its own (main, text encoder /
# anchor labelling, frozen-stage look-ahead, teacher) and RCCL adds its communicator streams: with the default, streams share queues
# and the overlap serialises -- measured 98.5 -> 90.5 views/s as soon as an RCCL process group exists, 95 with 8 queues (what is left
# is the gradient packing itself).  Must be in the environment before the HIP runtime initialises (the first torch.cuda call); a value set by the
# user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
