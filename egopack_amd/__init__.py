"""egopack_amd -- MI355X-native implementation of the EgoPack training hot path.

Layout: csrc/ (HIP kernels + C ABI, built into libegopack_hip.so), _lib.py (ctypes binding),
ops.py (autograd Functions over the C ABI), models/ (host-side mirror of the reference's
models.* interface), data.py (batches, edges, CSR, loaders), optim.py (flat-buffer Adam),
dist.py (one-process-per-GPU gradient exchange over RCCL), engine.py (train steps, hipGraph
capture), config.py (Hydra-compatible config loading / instantiate).
"""
__version__ = "0.1.0"
