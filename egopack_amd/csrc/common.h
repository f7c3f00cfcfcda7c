// Internal helpers shared by the gfx950 kernels of libegopack_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/egopack_hip.h"

namespace egk {

// ---- error plumbing ---------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define EGK_REQUIRE(cond, ...)               \
    do {                                     \
        if (!(cond)) {                       \
            ::egk::set_error(__VA_ARGS__);   \
            return EGK_EINVAL;               \
        }                                    \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ---- per-kernel profiling (HIP events on the launch stream) ------------------------------
enum KernelId {
    KID_GEMM_BF16_NN = 0,  // op(A) normal, op(B) normal  (forward linears)
    KID_GEMM_BF16_NT,      // B transposed in memory      (dX)
    KID_GEMM_BF16_TT,      // both transposed             (dW)
    KID_GEMM_BF16_TN,
    KID_GEMM_F32_NN,
    KID_GEMM_F32_NT,
    KID_GEMM_F32_TT,
    KID_GEMM_F32_TN,
    KID_GEMM_BF16_NN_G2,   // the same four layouts on the two-wave-group pipelined kernel (launches of <= 256 workgroups)
    KID_GEMM_BF16_NT_G2,
    KID_GEMM_BF16_TT_G2,
    KID_GEMM_BF16_TN_G2,
    KID_GEMM_BF16_NN_R96,  // 96-row tiles (row-major A): forward Linear / dX of outputs that 128-row tiles load unevenly
    KID_GEMM_BF16_NT_R96,
    KID_GEMM_BF16_NN_R64,  // 64-row tiles: launches of at most 128 tiles of 128 rows, one 4-wave workgroup per CU
    KID_GEMM_BF16_NT_R64,
    KID_GEMM_BF16_NN_R192, // 192 x 128 tiles, one 8-wave workgroup per CU on a 3-stage ring: outputs of ~256 such tiles (6144 x 1024)
    KID_GEMM_BF16_NT_R192,
    KID_GEMM_BF16_NN_T256, // 256 x 256 tiles, one 8-wave workgroup per CU: outputs of >= ~200 such tiles
    KID_GEMM_BF16_NT_T256,
    KID_GEMM_BF16_TT_T256,
    KID_GEMM_BF16_TN_T256,
    KID_GEMM_BF16_GROUP_NN,  // grouped launches (egk_gemm_grouped): several problems of one layout in one launch
    KID_GEMM_BF16_GROUP_NT,
    KID_GEMM_BF16_GROUP_TT,
    KID_GEMM_BF16_GENERIC, // bf16 MFMA on the register-staged kernel (K not a multiple of 64, unaligned rows, f32 storage)
    KID_GEMM_SPLITK_REDUCE,
    KID_COLSUM,
    KID_ROWLN_FWD,
    KID_ROWLN_BWD,
    KID_ROWLN_BWD_REDUCE,
    KID_GRAPHLN_STATS,
    KID_GRAPHLN_FWD,
    KID_GRAPHLN_BWD_STATS,
    KID_GRAPHLN_BWD,
    KID_GRAPHLN_BWD_REDUCE,
    KID_PE_ADD,
    KID_CSR_GATHER,
    KID_GATHER_MAX_FWD,
    KID_GATHER_MAX_BWD,
    KID_SEGMAX_FWD,
    KID_SEGMAX_BWD,
    KID_ROW_INV_NORM,
    KID_TOPK,
    KID_SCATTER_ADD_F64,
    KID_CE_FWD,
    KID_CE_BWD,
    KID_BCE_FWD,
    KID_BCE_BWD,
    KID_DROPOUT_FWD,
    KID_DROPOUT_BWD,
    KID_RELU_GATE,
    KID_CAST,
    KID_AXPBY,
    KID_SUM_SCALE,
    KID_ADAM,
    KID_COUNT
};

bool prof_on();
void prof_begin(int kid, hipStream_t s, double flops, double bytes);
void prof_end(hipStream_t s);

void set_adam_blocks(int n);       // loss_optim.hip: workgroup cap of the Adam launch (egk_tune key 6)
void set_zero_fill_blocks(int n);  // loss_optim.hip: workgroup cap of egk_zero_fill (egk_tune key 5)

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(int kid, hipStream_t s_, double flops, double bytes) : s(s_), on(prof_on()) {
        if (on) prof_begin(kid, s, flops, bytes);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};

// ---- device helpers ------------------------------------------------------------------------
constexpr int WAVE = 64;

// ---- row ownership shared by the launches of a chain ---------------------------------------------------------------------------
// The hardware deals consecutive workgroups round-robin over the 8 XCDs (blockIdx % 8 = XCD; MI355X_MICROARCH.md, workgroup
// dispatch) and every XCD has its own L2.  A tensor that one launch writes and the next one reads is served from the READER's L2
// when the same XCD wrote those rows, and across the fabric otherwise -- measured (tools/exp/xcd_affinity.hip, a [6144, 1024] bf16
// tensor read and another written per launch): 4.0-4.4 us when producer and consumer agree on who owns a row, 8.3-11.2 us when they
// do not.  So every row kernel of the step gives XCD x the CONTIGUOUS rows [x * per, (x + 1) * per), per = ceil(rows / 8) rounded up
// to whole workgroup passes -- the rows whose output tiles the contraction kernels hand to XCD x (gemm.hip: group_m = tiles_m / 8)
// -- instead of dealing the rows round-robin with the workgroups.  Placement is a speed matter only: which workgroup computes a row
// never changes the row's result.  Launches without whole rounds of 8 workgroups keep the round-robin walk.
struct RowWalk {
    int first, end, step;  // this wave's rows: first, first + step, first + 2 * step, ... < end
};
__device__ __forceinline__ RowWalk row_walk(int blk, int nblk, int row0, int rows, int wave, int wpb) {
    if (nblk >= 8 && (nblk & 7) == 0) {
        const int xcd = blk & 7, slot = blk >> 3, nslot = nblk >> 3;
        const int per = (((rows - row0 + 7) >> 3) + wpb - 1) / wpb * wpb;
        const int b = row0 + xcd * per;
        const int e = b + per < rows ? b + per : rows;
        return RowWalk{b + slot * wpb + wave, e, nslot * wpb};
    }
    return RowWalk{row0 + blk * wpb + wave, rows, nblk * wpb};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Philox4x32-10 (Salmon et al. 2011).  counter = (ctr_lo, ctr_hi, 0, 0), key = seed.
__device__ __forceinline__ uint4 philox4x32_10(uint64_t ctr, uint64_t seed) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// uniform in [0,1) from 32 random bits (24-bit mantissa)
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- activation element types: float or bf16 (stored as unsigned short), always computed in f32 -------------
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// 4 consecutive elements starting at column c of a row (zero beyond cols); vec = aligned wide access allowed
__device__ __forceinline__ float4 ld4t(const float* __restrict__ p, int c, int cols, bool vec) {
    if (vec && c + 4 <= cols) return *reinterpret_cast<const float4*>(p + c);
    float4 v;
    v.x = c + 0 < cols ? p[c + 0] : 0.f;
    v.y = c + 1 < cols ? p[c + 1] : 0.f;
    v.z = c + 2 < cols ? p[c + 2] : 0.f;
    v.w = c + 3 < cols ? p[c + 3] : 0.f;
    return v;
}
__device__ __forceinline__ float4 ld4t(const bf16_t* __restrict__ p, int c, int cols, bool vec) {
    float4 v;
    if (vec && c + 4 <= cols) {
        const uint2 r = *reinterpret_cast<const uint2*>(p + c);
        v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
        v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
        return v;
    }
    v.x = c + 0 < cols ? bf2f(p[c + 0]) : 0.f;
    v.y = c + 1 < cols ? bf2f(p[c + 1]) : 0.f;
    v.z = c + 2 < cols ? bf2f(p[c + 2]) : 0.f;
    v.w = c + 3 < cols ? bf2f(p[c + 3]) : 0.f;
    return v;
}
__device__ __forceinline__ void st4t(float* __restrict__ p, int c, int cols, bool vec, float4 v) {
    if (vec && c + 4 <= cols) {
        *reinterpret_cast<float4*>(p + c) = v;
        return;
    }
    if (c + 0 < cols) p[c + 0] = v.x;
    if (c + 1 < cols) p[c + 1] = v.y;
    if (c + 2 < cols) p[c + 2] = v.z;
    if (c + 3 < cols) p[c + 3] = v.w;
}
__device__ __forceinline__ void st4t(bf16_t* __restrict__ p, int c, int cols, bool vec, float4 v) {
    if (vec && c + 4 <= cols) {
        uint2 pk;
        pk.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
        pk.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
        *reinterpret_cast<uint2*>(p + c) = pk;
        return;
    }
    if (c + 0 < cols) p[c + 0] = f2bf(v.x);
    if (c + 1 < cols) p[c + 1] = f2bf(v.y);
    if (c + 2 < cols) p[c + 2] = f2bf(v.z);
    if (c + 3 < cols) p[c + 3] = f2bf(v.w);
}
__device__ __forceinline__ float ld1t(const float* p) { return *p; }
__device__ __forceinline__ float ld1t(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void st1t(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1t(bf16_t* p, float v) { *p = f2bf(v); }

// ---- split tee: the bf16 halves (hi = bf16(x), lo = bf16(x - hi): egk_split_bf16's arithmetic) of an f32 result, stored by
// the kernel that produces it (egk_tee_split_next).  Row r of the result goes to hi / lo + r * ld.
struct SplitTee {
    bf16_t* hi;
    bf16_t* lo;
    long long ld;
};
__device__ __forceinline__ void tee4(const SplitTee& t, long long row, int c, int cols, bool vec, const float4& v) {
    if (!t.lo) return;
    const float x[4] = {v.x, v.y, v.z, v.w};
    bf16_t h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k] = f2bf(x[k]);
        l[k] = f2bf(x[k] - bf2f(h[k]));
    }
    bf16_t* hp = t.hi + row * t.ld + c;
    bf16_t* lp = t.lo + row * t.ld + c;
    if (vec && c + 4 <= cols && (t.ld & 3) == 0) {
        *reinterpret_cast<uint2*>(hp) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        *reinterpret_cast<uint2*>(lp) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
    } else {
        for (int k = 0; k < 4 && c + k < cols; ++k) {
            hp[k] = h[k];
            lp[k] = l[k];
        }
    }
}
// the tee armed by egk_tee_split_next, handed to (and cleared by) the next row-kernel launcher; null pointers when none is armed
SplitTee take_split_tee();
bool split_tee_armed();

// One-shot "slab input" of a row kernel (egk_slab_input_next): its f32 input is given as the two K slabs of a split contraction whose
// reduce launch was left out (egk_gemm_defer_reduce_next): element (r, c) = (x[r, c] + x2[r, c]) + bias[c] -- gemm_splitk_reduce's
// arithmetic, so the same bits -- and the kernel also stores that value to x_out[r, c] (row stride = cols), so that later readers
// find the reduced matrix.  x2 == nullptr: a plain input.
struct SlabInput {
    const float* x2;
    const float* bias;
    float* x_out;
};
SlabInput take_slab_input();
bool slab_input_armed();
bool take_defer_reduce();  // egk_gemm_defer_reduce_next: the next split-K egk_gemm leaves its slabs unreduced
void arm_split_tee(bf16_t* hi, bf16_t* lo, long long ld);
void arm_slab_input(const float* x2, const float* bias, float* x_out);
void arm_defer_reduce(bool on);

// ---- Adam on one element (torch.optim.Adam single-tensor formulas, L2 weight decay): the ONE definition the optimizer kernels and the
// weight-gradient epilogue share, so that a parameter stepped in either place gets the same bits --------------------------------------
struct AdamConsts {
    float step, bc2s, gs, b1, b2, eps, wd;  // step = lr / (1 - b1^t), bc2s = sqrt(1 - b2^t), gs = gradient scale
};
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdamConsts& c) {
#pragma clang fp contract(off)  // (no fused multiply-adds: which products the compiler fuses depends on the code around the inlined body)
    const float gg = g * c.gs + c.wd * p;
    m = m + (gg - m) * (1.f - c.b1);          // exp_avg.lerp_(grad, 1 - beta1)
    v = v * c.b2 + (1.f - c.b2) * gg * gg;    // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / c.bc2s + c.eps;
    p = p - c.step * (m / denom);
}
// host-side dispatch on an EGK_F32 / EGK_BF16 activation type: ``using T = ...`` inside CALL
#define EGK_DISPATCH_T(dtype, ...)                                               \
    do {                                                                         \
        if ((dtype) == EGK_BF16) { using T = ::egk::bf16_t; __VA_ARGS__; }       \
        else if ((dtype) == EGK_F32) { using T = float; __VA_ARGS__; }           \
        else { ::egk::set_error("unknown activation dtype %d", (int)(dtype)); return EGK_EINVAL; } \
    } while (0)

}  // namespace egk
