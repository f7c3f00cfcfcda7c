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
    KID_AXPBY,
    KID_SUM_SCALE,
    KID_ADAM,
    KID_COUNT
};

bool prof_on();
void prof_begin(int kid, hipStream_t s, double flops, double bytes);
void prof_end(hipStream_t s);

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

}  // namespace egk
