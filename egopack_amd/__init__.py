"""egopack_amd -- MI355X-native implementation of the EgoPack training hot path.

Layout: csrc/ (HIP kernels + C ABI, built into libegopack_hip.so), _lib.py (ctypes binding),
ops.py (autograd Functions over the C ABI), models/ (host-side mirror of the reference's
models.* interface), data.py (batches, edges, CSR, loaders), optim.py (flat-buffer Adam),
dist.py (one-process-per-GPU gradient exchange over RCCL), engine.py (train steps, hipGraph
capture), config.py (Hydra-compatible config loading / instantiate).
"""
import os as _os


def tune_single_process_runtime(parallel_heads: int = 3) -> bool:
    """Call FIRST in a single-process entry point (before anything initialises the HIP device; never in a process that
    will create a torch.distributed / RCCL group): caps the runtime's hardware queues at 3 for steps of up to three
    parallel task heads.

    A captured single-GPU step forks a handful of streams (the task heads, weight gradients, Adam); every cross-queue
    hand-off on its critical chain costs about 10 us, and with 3 queues instead of the default 4 the MI355X runs the
    three-task step 1.3 % faster (1.781 vs 1.804 ms, same box; 5 and more collapse: 3.9 ms).  Four heads want the fourth
    queue (4-task step 2.26 vs 2.20 ms under 3), and with RCCL in the process the communication streams need queues of
    their own (exchange path 2.05 vs 1.96 ms under 3, and a captured step with an RCCL group crashed under 3): those keep
    the runtime default.  An explicit GPU_MAX_HW_QUEUES in the environment always wins."""
    if int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1 or parallel_heads > 3:
        return False
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
    return True


__version__ = "0.1.0"
