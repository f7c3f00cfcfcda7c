#!/usr/bin/env python3
"""Input side of one bench step (6144 nodes x 3 segments x 1536 features): gather from the device-resident feature
store vs the host -> device copy of the same block from pinned memory (what the reference's loader path costs)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import feature_store as FS

N, S, F = 6144, 3, 1536
rows = 1 << 20  # 1 M rows x 1536 bf16 = 3.2 GB resident
table = torch.randn(rows, F, device="cuda").to(torch.bfloat16)
store = FS.FeatureStore.__new__(FS.FeatureStore)
store.table, store.rows, store.features_size, store.offsets = table, rows, F, {}
g = torch.Generator().manual_seed(0)
# windows of consecutive frames, as the datasets produce: 3 nearby rows per node, nodes of a sequence close together
base = torch.randint(0, rows - 64, (N // 32, 1, 1), generator=g) + torch.arange(32).view(1, 32, 1) * 2
idx = (base + torch.randint(0, 8, (N // 32, 32, S), generator=g)).view(N, S)
idx_dev = idx.cuda()
out = torch.empty(N, S, F, device="cuda", dtype=torch.bfloat16)
nbytes = out.numel() * 2
us = time_us(lambda: store.gather(idx_dev, out=out), 20)
print(f"gather from resident store   {us:8.1f} us  {2 * nbytes / us / 1e3:7.0f} GB/s (read + write {2 * nbytes / 1e6:.1f} MB)")
idx_rand = torch.randint(0, rows, (N, S), generator=g).cuda()
us = time_us(lambda: store.gather(idx_rand, out=out), 20)
print(f"  ... uniformly random rows  {us:8.1f} us  {2 * nbytes / us / 1e3:7.0f} GB/s")
host = torch.empty(N, S, F, dtype=torch.bfloat16).pin_memory()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    out.copy_(host, non_blocking=True)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    out.copy_(host, non_blocking=True)
e1.record()
torch.cuda.synchronize()
us_h2d = e0.elapsed_time(e1) * 1e3 / 10
print(f"pinned host -> device copy   {us_h2d:8.1f} us  {nbytes / us_h2d / 1e3:7.1f} GB/s ({nbytes / 1e6:.1f} MB bf16; f32 features: x2)")
pin_idx = idx.pin_memory()
e0.record()
for _ in range(10):
    idx_dev.copy_(pin_idx, non_blocking=True)
e1.record()
torch.cuda.synchronize()
print(f"index matrix host -> device  {e0.elapsed_time(e1) * 1e3 / 10:8.1f} us  ({idx.numel() * 8 / 1e3:.0f} KB)")
