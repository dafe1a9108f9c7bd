"""coin_amd: MI355X-native implementation of COIN's adaptation-training hot path.

Only what the path needs: ``csrc/`` (HIP kernels + C ABI, ``include/coin_hip.h``), ``kernels``
(tensor-level binding) and the host-side mirror of the reference's registries / modules.
"""
__version__ = "0.1.0"
