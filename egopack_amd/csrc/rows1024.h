// Helpers for row kernels on bf16 activations with rows of EXACTLY 1024 columns (H = 1024), used by the banded mean gather and
// the positional-encoding add of graph_ops.hip:
//   * lane l of the wave that owns a row holds columns [8l, 8l + 8) and [512 + 8l, 512 + 8l + 8): two 16-byte accesses per
//     row, each a 1-KiB coalesced wave-instruction, the row as 8 packed registers until it is used;
//   * ALL the rows a wave will touch in a sweep are requested before the first byte is used.
// Measured (tools/row_bench.py, [6144, 1024] bf16): banded gather 8.5 -> 5.6 us stand-alone, PE add 8.9 -> 7.3 us inside the step.
// The same treatment of the LayerNorm kernels (affine rows staged in LDS, rows requested before the statistics prologue) was
// built and measured in round 4 and NOT kept: stand-alone 7.4 -> 7.5 us (row LN forward), 26.9 -> 29.5 us (forward +
// backward), inside the step 12.1 -> 14.1 us -- those launches are bound by where their input comes from (the tensor the
// previous launch just wrote, read across XCDs) and by the launch boundary, not by how many loads a wave keeps in flight.
#pragma once
#include "common.h"

namespace egk {
namespace r1k {

constexpr int COLS = 1024;

struct Raw {
    uint4 c0, c1;  // 8 + 8 packed bf16
};

__device__ __forceinline__ Raw ld_raw(const bf16_t* __restrict__ row, int lane) {
    Raw r;
    r.c0 = *reinterpret_cast<const uint4*>(row + 8 * lane);
    r.c1 = *reinterpret_cast<const uint4*>(row + 512 + 8 * lane);
    return r;
}
__device__ __forceinline__ void st_raw(bf16_t* __restrict__ row, int lane, const Raw& r) {
    *reinterpret_cast<uint4*>(row + 8 * lane) = r.c0;
    *reinterpret_cast<uint4*>(row + 512 + 8 * lane) = r.c1;
}
__device__ __forceinline__ void unpack(const Raw& r, float (&f)[16]) {
    const unsigned w[8] = {r.c0.x, r.c0.y, r.c0.z, r.c0.w, r.c1.x, r.c1.y, r.c1.z, r.c1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ Raw pack(const float (&f)[16]) {
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
    Raw r;
    r.c0 = make_uint4(w[0], w[1], w[2], w[3]);
    r.c1 = make_uint4(w[4], w[5], w[6], w[7]);
    return r;
}
// column of element j (0 .. 15) of lane ``lane``
__device__ __forceinline__ int col_of(int lane, int j) { return (j < 8 ? 0 : 512) + 8 * lane + (j & 7); }

// f32 vector of 1024 entries (a row of the positional-encoding table) -> the 16 values of this lane
__device__ __forceinline__ void ld_vec16(const float* __restrict__ v, int lane, float (&f)[16]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 a = *reinterpret_cast<const float4*>(v + h * 512 + 8 * lane);
        const float4 b = *reinterpret_cast<const float4*>(v + h * 512 + 8 * lane + 4);
        f[8 * h + 0] = a.x; f[8 * h + 1] = a.y; f[8 * h + 2] = a.z; f[8 * h + 3] = a.w;
        f[8 * h + 4] = b.x; f[8 * h + 5] = b.y; f[8 * h + 6] = b.z; f[8 * h + 7] = b.w;
    }
}
}  // namespace r1k
}  // namespace egk
