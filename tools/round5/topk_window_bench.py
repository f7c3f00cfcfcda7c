"""Where the window search's time goes: the launch alone (dot product precomputed), varying rows, bank size and how many candidates
the window admits (bf16-exact operands: E is minimal, k candidates per row)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopack_amd import ops

DEV = "cuda"


def run(N, K, H, k, exact, reps=30):
    g = torch.Generator().manual_seed(N + K)
    f, bank = torch.randn(N, H, generator=g), torch.randn(K, H, generator=g)
    if exact:
        f, bank = f.bfloat16().float(), bank.bfloat16().float()
    f, bank = f.to(DEV), bank.to(DEV)
    cand = torch.zeros(N, dtype=torch.int32, device=DEV)
    with ops.compute_mode("bf16"):
        bn = ops.row_inv_norm(bank)
        ops._window_stats["cand"] = cand
        for _ in range(3):
            ops.nearest_prototypes(f, bank, k, "cosine", bn)
        ops._window_stats["cand"] = None
        torch.cuda.synchronize()
        ops.prof_enable(True) if hasattr(ops, "prof_enable") else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.nearest_prototypes(f, bank, k, "cosine", bn)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, float(cand.float().mean())


if __name__ == "__main__":
    for N, K, H, k, exact in [(2048, 4096, 1024, 4, False), (2048, 4096, 1024, 4, True), (6144, 4096, 1024, 4, False),
                              (6144, 4096, 1024, 4, True), (2048, 1024, 1024, 4, False), (2048, 4096, 256, 4, False)]:
        us, c = run(N, K, H, k, exact)
        print(f"N={N} K={K} H={H} k={k} exact_bf16={exact}: whole search {us:.1f} us/call, candidates/row {c:.1f}", flush=True)
