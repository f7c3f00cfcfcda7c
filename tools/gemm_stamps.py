#!/usr/bin/env python3
"""Diagnostic build (NOT the product library): the 256 x 256 contraction kernel with s_memtime stamps around the sections
of its K loop, to see where a wave's cycles go.  Builds tools/exp/build/libegopack_stamps.so with -DEGK_GEMM_STAMPS (run
with --build here, where hipcc is; the .so travels to the GPU box), then launches one shape and prints the median wave.
Usage: python tools/gemm_stamps.py --build | python tools/gemm_stamps.py M N K [tA tB]"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
OUT = REPO / "tools" / "exp" / "build"
LIB = OUT / "libegopack_stamps.so"

if len(sys.argv) > 1 and sys.argv[1] == "--build":
    from egopack_amd import build as B
    OUT.mkdir(parents=True, exist_ok=True)
    objs = []
    for src in B.sources():
        obj = OUT / (src.stem + ".o")
        subprocess.run([B._hipcc(), *B.FLAGS, "-DEGK_GEMM_STAMPS", "-c", str(src), "-o", str(obj)], check=True)
        objs.append(str(obj))
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", str(LIB), *objs], check=True)
    print(LIB)
    sys.exit(0)

import torch

from egopack_amd import _lib

_lib.LIB_PATH = LIB
from egopack_amd import ops

if "--phases" in sys.argv:
    # python tools/gemm_stamps.py --phases M N K tA tB splitk [variant]: entry -> first tile landed -> K loop -> epilogue,
    # per workgroup, of the 128-row pipelined kernels (the launch the policy would make unless a variant is forced)
    a = [v for v in sys.argv[1:] if not v.startswith("--")]
    M, N, K, tA, tB, sk = (int(v) for v in a[:6])
    lib = _lib.load()
    lib.egk_gemm_set_pipeline(int(a[6]) if len(a) > 6 else 1)
    bf = torch.bfloat16
    A = torch.randn((K, M) if tA else (M, K), device="cuda").to(bf)
    B = torch.randn((K, N) if tB else (N, K), device="cuda").to(bf)
    acc = bool(tA and tB)
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if acc else bf)
    slab = sk * M * N * 4 if sk > 1 else 0
    nwg_max = ((M + 63) // 64) * ((N + 127) // 128) * sk
    ws = torch.zeros(slab + nwg_max * 64 + 64, dtype=torch.uint8, device="cuda")
    d = _lib.GemmDesc()
    d.M, d.N, d.K1, d.K2 = M, N, K, 0
    d.A1, d.B1 = A.data_ptr(), B.data_ptr()
    d.lda1, d.ldb1 = A.shape[1], B.shape[1]
    d.transA, d.transB = tA, tB
    d.a_dtype = d.b_dtype = ops.BF16
    d.c_dtype, d.compute = (ops.F32 if acc else ops.BF16), ops.BF16
    d.C, d.ldc, d.alpha, d.splitk, d.accumulate = out.data_ptr(), N, 1.0, sk, int(acc)
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        assert lib.egk_gemm(ops._stream(), C.byref(d)) == 0
    torch.cuda.synchronize()
    ws[slab:].zero_()
    torch.cuda.synchronize()
    e0.record()
    assert lib.egk_gemm(ops._stream(), C.byref(d)) == 0
    e1.record()
    torch.cuda.synchronize()
    st = ws[slab: slab + nwg_max * 64].view(torch.int64).view(-1, 8).cpu().double()
    st = st[st[:, 3] > 0]
    clk = (st[:, 3] / st[:, 4]).median().item() * 0.1
    t0 = st[:, 5].min()
    start_us, end_us = (st[:, 5] - t0) / 100, (st[:, 5] - t0 + st[:, 4]) / 100
    med = st.median(0).values
    print(f"{M}x{N}x{K} tA={tA} tB={tB} splitk={sk}: {st.shape[0]} workgroups x {int(med[6])} K tiles (per wave group); events around the "
          f"launch (incl. the slab reduce if any): {e0.elapsed_time(e1) * 1e3:.1f} us; in-kernel clock {clk:.2f} GHz")
    print(f"  first workgroup entry -> last workgroup exit: {end_us.max():.1f} us; workgroup entries spread over {start_us.max():.1f} us "
          f"(median {start_us.median():.1f})")
    for name, i in (("entry -> first K tile landed", 0), ("K loop", 1), ("epilogue: until the stores are issued", 7),
                    ("epilogue: until they are acknowledged", 2), ("workgroup lifetime", 3)):
        print(f"  {name:42s} median {med[i] / clk / 1e3:6.2f} us   max {st[:, i].max().item() / clk / 1e3:6.2f} us")
    sys.exit(0)

M, N, K = (int(v) for v in sys.argv[1:4])
tA, tB = (int(v) for v in sys.argv[4:6]) if len(sys.argv) > 5 else (0, 0)
lib = _lib.load()
lib.egk_gemm_set_pipeline(7)
bf = torch.bfloat16
A = torch.randn((K, M) if tA else (M, K), device="cuda").to(bf)
B = torch.randn((K, N) if tB else (N, K), device="cuda").to(bf)
out = torch.empty(M, N, device="cuda", dtype=bf)
tiles = ((M + 255) // 256) * ((N + 255) // 256)
ws = torch.zeros(tiles * 8 * 8, dtype=torch.int64, device="cuda")
d = _lib.GemmDesc()
d.M, d.N, d.K1, d.K2 = M, N, K, 0
d.A1, d.B1 = A.data_ptr(), B.data_ptr()
d.lda1, d.ldb1 = A.shape[1], B.shape[1]
d.transA, d.transB = tA, tB
d.a_dtype = d.b_dtype = ops.BF16
d.c_dtype, d.compute = ops.BF16, ops.BF16
d.C, d.ldc, d.alpha, d.splitk = out.data_ptr(), N, 1.0, 1
d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 8
for _ in range(20):  # back-to-back launches: the clocks settle
    assert lib.egk_gemm(ops._stream(), C.byref(d)) == 0
torch.cuda.synchronize()
st = ws.view(tiles * 8, 8).cpu().double()
med = st.median(0).values
nt = med[6].item()
wait, issue, head, body, total, real = (med[i].item() for i in range(6))
print(f"{M}x{N}x{K} tA={tA} tB={tB}: {tiles} workgroups, {int(nt)} K tiles per wave; median wave:")
print(f"  in-kernel clock {total / real * 0.1:.2f} GHz, kernel body {real / 100:.1f} us")
for name, v in (("vmcnt(0) + barrier", wait), ("DMA issue (8 pieces)", issue), ("first fragment reads -> ready", head),
                ("4 MFMA phases (64 MFMAs = 1024 pipe cycles; 2 waves share a SIMD)", body)):
    print(f"  {name:70s} {v / nt:8.0f} cycles / K tile  ({100 * v / total:4.1f} % of the wave)")
print(f"  loop total {(wait + issue + head + body) / nt:8.0f} cycles / K tile; prologue + epilogue {100 * (total - wait - issue - head - body) / total:.1f} %")
if "--raw" in sys.argv:
    torch.set_printoptions(linewidth=200, sci_mode=False)
    print("rows = waves 0..7 of workgroup 0 and of the last workgroup; columns = wait issue head body total real nt")
    print(st[:8, :7].long())
    print(st[-8:, :7].long())
    acc = st[:, :4].sum(1)
    print("sections / total: min %.3f median %.3f max %.3f" % ((acc / st[:, 4]).min(), (acc / st[:, 4]).median(), (acc / st[:, 4]).max()))
