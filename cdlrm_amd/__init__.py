"""cdlrm_amd -- MI355X-native look-ahead embedding-cache DLRM training path (drop-in for lkp411/cDLRM's
cache_manager.py / model_no_ddp.py / main_no_ddp.py surface) over hand-written gfx950 HIP kernels.

Compute lives in csrc/libcdlrm_hip.so (C ABI: include/cdlrm_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
