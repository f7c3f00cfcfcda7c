#!/usr/bin/env python3
"""dW-form contractions (both operands k-major, K = all nodes) on the deep sub-staged ring (egk_gemm_set_pipeline(804 / 805))
against the 2-stage 128 x 128 kernel: bit-equality, then device time per launch -- single launches and grouped launches,
with the operands hot (one set, re-read) and rotating over enough sets that they come from HBM (> 256 MiB in all)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
dev = "cuda"
bf = torch.bfloat16


def mk(K, M, N, sets):
    return ([torch.randn(K, M, device=dev).to(bf) for _ in range(sets)], [torch.randn(K, N, device=dev).to(bf) for _ in range(sets)])


def check():
    ok = True
    for (M, N, K, bias) in [(1024, 1024, 6144, True), (1024, 4608, 2048, True), (640, 1024, 2048, True), (1024, 1024, 192, False),
                            (115, 1024, 2048, True), (1024, 1024, 64, True)]:
        A = torch.randn(K, (M + 7) // 8 * 8, device=dev).to(bf)
        B = torch.randn(K, N, device=dev).to(bf)
        res = {}
        for knob in (800, 804, 805):
            lib.egk_gemm_set_pipeline(knob)
            lib.egk_gemm_set_pipeline(3)
            out = torch.full((M, N), 0.5, device=dev)
            db = torch.full((M,), 0.25, device=dev)
            ops.gemm(M, N, A, A.shape[1], B, N, K, out, N, transA=True, transB=True, accumulate=True, compute=ops.BF16,
                     dbias=db if bias else None, allow_splitk=False)
            torch.cuda.synchronize()
            res[knob] = (out.clone(), db.clone())
        lib.egk_gemm_set_pipeline(1)
        lib.egk_gemm_set_pipeline(800)
        ref = (A.float().t()[:M] @ B.float()) + 0.5
        e = (res[800][0] - ref).abs().max().item()
        for knob in (804, 805):
            same = torch.equal(res[knob][0], res[800][0]) and torch.equal(res[knob][1], res[800][1])
            ok = ok and same
            print(f"check M={M} N={N} K={K} bias={bias} knob={knob}: bit-equal={same}  (2-stage kernel vs f32 matmul: {e:.3e})")
    # grouped
    H, Kn = 1024, 6144
    gs = [torch.randn(Kn, H, device=dev).to(bf) for _ in range(6)]
    xs = [torch.randn(Kn, H, device=dev).to(bf) for _ in range(6)]
    res = {}
    for knob in (800, 804, 805):
        lib.egk_gemm_set_pipeline(knob)
        outs = [torch.full((H, H), 0.5, device=dev) for _ in range(6)]
        dbs = [torch.zeros(H, device=dev) for _ in range(6)]
        probs = [((H, H, gs[i], H, xs[i], H, Kn, outs[i], H), dict(transA=True, transB=True, accumulate=True, compute=ops.BF16, dbias=dbs[i]))
                 for i in range(6)]
        ops.gemm_grouped(probs, four_wave=True)
        torch.cuda.synchronize()
        res[knob] = torch.stack(outs + [d.expand(H, H) for d in dbs]).clone()
    lib.egk_gemm_set_pipeline(800)
    for knob in (804, 805):
        same = torch.equal(res[knob], res[800])
        ok = ok and same
        print(f"check grouped x6 knob={knob}: bit-equal={same}")
    return ok


def bench():
    H = 1024
    for name, M, N, K in [("dW TRN1 merged", 1024, 4608, 6144), ("dW HxH", 1024, 1024, 6144), ("dW TRN1 T=256", 1024, 4608, 16384),
                          ("dW Hp4096 TRN2", 4096, 4096, 6144), ("dW Hp4096 TRN1", 4096, 4608, 6144)]:
        sets = max(2, int(400e6 // (2.0 * K * (M + N))) + 1)
        As, Bs = mk(K, M, N, sets)
        out = torch.zeros(M, N, device=dev)
        fl = 2.0 * M * N * K
        row = []
        for label, knobs in (("v3", (800, 3)), ("v6", (800, 6)), ("v13", (800, 13)), ("sub4", (804, 3)), ("sub5", (805, 3)), ("policy", (800, 1))):
            for k in knobs:
                lib.egk_gemm_set_pipeline(k)
            hot = time_us(lambda: ops.gemm(M, N, As[0], M, Bs[0], N, K, out, N, transA=True, transB=True, accumulate=True, compute=ops.BF16,
                                           allow_splitk=(label == "policy")), 20)
            cnt = [0]

            def rot():
                i = cnt[0] % sets
                cnt[0] += 1
                ops.gemm(M, N, As[i], M, Bs[i], N, K, out, N, transA=True, transB=True, accumulate=True, compute=ops.BF16,
                         allow_splitk=(label == "policy"))
            cold = time_us(rot, 4 * sets)
            row.append(f"{label}: {hot:6.1f}/{cold:6.1f} us ({fl / cold / 1e6:5.0f} TF/s)")
        lib.egk_gemm_set_pipeline(800)
        lib.egk_gemm_set_pipeline(1)
        print(f"{name:16s} {M}x{N}x{K} sets={sets}  hot/cold  " + "  ".join(row), flush=True)
    # grouped H x H weight gradients, 6 and 8 problems, and the step's tail group (TRN dW1 + dW2 + dW3)
    K = 6144
    for n in (6, 8):
        sets = 3
        gs = [[torch.randn(K, H, device=dev).to(bf) for _ in range(n)] for _ in range(sets)]
        xs = [[torch.randn(K, H, device=dev).to(bf) for _ in range(n)] for _ in range(sets)]
        outs = [torch.zeros(H, H, device=dev) for _ in range(n)]
        dbs = [torch.zeros(H, device=dev) for _ in range(n)]
        fl = n * 2.0 * H * H * K
        row = []
        for label, knob in (("2-stage", 800), ("sub4", 804), ("sub5", 805)):
            lib.egk_gemm_set_pipeline(knob)
            cnt = [0]

            def rot():
                s_ = cnt[0] % sets
                cnt[0] += 1
                ops.gemm_grouped([((H, H, gs[s_][i], H, xs[s_][i], H, K, outs[i], H),
                                   dict(transA=True, transB=True, accumulate=True, compute=ops.BF16, dbias=dbs[i])) for i in range(n)], four_wave=True)
            us = time_us(rot, 12)
            row.append(f"{label}: {us:6.1f} us ({fl / us / 1e6:5.0f} TF/s)")
        lib.egk_gemm_set_pipeline(800)
        print(f"grouped x{n} HxH K={K} (rotating {sets} sets)  " + "  ".join(row), flush=True)
    sets = 4
    x = [torch.randn(K, 4608, device=dev).to(bf) for _ in range(sets)]
    dy = [[torch.randn(K, H, device=dev).to(bf) for _ in range(3)] for _ in range(sets)]
    hh = [[torch.randn(K, H, device=dev).to(bf) for _ in range(2)] for _ in range(sets)]
    o1 = torch.zeros(H, 4608, device=dev)
    o2, o3 = torch.zeros(H, H, device=dev), torch.zeros(H, H, device=dev)
    dbs = [torch.zeros(H, device=dev) for _ in range(3)]
    fl = 2.0 * K * H * (4608 + 2 * H)
    row = []
    for label, knob in (("2-stage", 800), ("sub4", 804), ("sub5", 805)):
        lib.egk_gemm_set_pipeline(knob)
        cnt = [0]

        def rot():
            s_ = cnt[0] % sets
            cnt[0] += 1
            kw = dict(transA=True, transB=True, accumulate=True, compute=ops.BF16)
            ops.gemm_grouped([((H, 4608, dy[s_][0], H, x[s_], 4608, K, o1, 4608), dict(dbias=dbs[0], **kw)),
                              ((H, H, dy[s_][1], H, hh[s_][0], H, K, o2, H), dict(dbias=dbs[1], **kw)),
                              ((H, H, dy[s_][2], H, hh[s_][1], H, K, o3, H), dict(dbias=dbs[2], **kw))], four_wave=True)
        us = time_us(rot, 12)
        row.append(f"{label}: {us:6.1f} us ({fl / us / 1e6:5.0f} TF/s)")
    lib.egk_gemm_set_pipeline(800)
    print(f"tail group (TRN dW1 + dW2 + dW3), K={K} (rotating {sets} sets)  " + "  ".join(row), flush=True)


if __name__ == "__main__":
    ok = check()
    print("BIT-EQUAL" if ok else "MISMATCH", flush=True)
    bench()
