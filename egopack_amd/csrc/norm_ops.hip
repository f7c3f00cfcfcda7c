// Normalisation kernels: per-row LayerNorm(+ReLU+dropout), graph-mode LayerNorm(+LeakyReLU)
// with per-segment global statistics, and column sums (bias gradients).
//
// Common shape: ONE WAVE PER ROW, 4 waves per workgroup, the row held in registers as NV float4
// per lane (column of element t of chunk i of lane l = (i*64 + l)*4 + t), so that
//   * every HBM access is a 1-KiB coalesced wave-instruction (16 B per lane),
//   * row reductions are wavefront shuffles (no LDS, no barrier),
//   * column reductions (dw/db) accumulate in registers across the rows a wave walks, because the
//     lane->column map is the same for every row, and are finished by a small second launch that
//     sums the per-workgroup partials in a fixed order (bitwise reproducible, no atomics).
// All of these are HBM-bound: algorithmic bytes are listed per kernel in DESIGN.md.
#include "common.h"

namespace egk {

constexpr int WPB = 4;  // waves (rows in flight) per workgroup

template <int NV>
struct Row {
    float4 v[NV];
};

template <int NV, typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ p, int cols, bool vec, int lane, Row<NV>& r) {
#pragma unroll
    for (int i = 0; i < NV; ++i) r.v[i] = ld4t(p, (i * 64 + lane) * 4, cols, vec);
}
template <int NV, typename T>
__device__ __forceinline__ void store_row(T* __restrict__ p, int cols, bool vec, int lane, const Row<NV>& r) {
#pragma unroll
    for (int i = 0; i < NV; ++i) st4t(p, (i * 64 + lane) * 4, cols, vec, r.v[i]);
}
__device__ __forceinline__ float& el(float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }
__device__ __forceinline__ float el(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

// -------------------------------------------------------------------------------------------
// row LayerNorm (+ReLU, +dropout)
// -------------------------------------------------------------------------------------------
// rows [row_begin, rows) are walked by ``nblk`` workgroups, this one being number ``blk`` (the plain launch: 0, all rows,
// gridDim.x, blockIdx.x; the grouped launch: one row range and one (w, b) pair per blockIdx.y)
template <int NV, typename T, bool FULL>
__device__ __forceinline__ void rowln_fwd_body(const T* __restrict__ x, const float* __restrict__ w,
                                               const float* __restrict__ b, T* __restrict__ y,
                                               float* __restrict__ mean, float* __restrict__ rstd,
                                               uint8_t* __restrict__ mask, int row_begin, int rows, int cols, float eps, int relu,
                                               float p, uint64_t seed, uint64_t offset,
                                               const uint64_t* __restrict__ dev_offset, int blk, int nblk,
                                               const SplitTee tee = SplitTee{nullptr, nullptr, 0},
                                               const SlabInput si = SlabInput{nullptr, nullptr, nullptr}) {
    if (dev_offset) offset += dev_offset[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (FULL) cols = NV * 256;  // exact-width rows: every bounds check below folds away
    const bool vec = FULL || (cols & 3) == 0;
    Row<NV> wv, bv;
    load_row<NV>(w, cols, vec, lane, wv);
    load_row<NV>(b, cols, vec, lane, bv);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const RowWalk rw = row_walk(blk, nblk, row_begin, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        Row<NV> r;
        load_row<NV>(x + (long long)row * cols, cols, vec, lane, r);
        if constexpr (sizeof(T) == 4) {
            if (si.x2) {  // the input as two K slabs + bias (egk_slab_input_next): gemm_splitk_reduce's arithmetic, stored for later readers
                Row<NV> r2;
                load_row<NV>(si.x2 + (long long)row * cols, cols, vec, lane, r2);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    r.v[i].x += r2.v[i].x; r.v[i].y += r2.v[i].y; r.v[i].z += r2.v[i].z; r.v[i].w += r2.v[i].w;
                }
                if (si.bias) {
                    load_row<NV>(si.bias, cols, vec, lane, r2);
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        r.v[i].x += r2.v[i].x; r.v[i].y += r2.v[i].y; r.v[i].z += r2.v[i].z; r.v[i].w += r2.v[i].w;
                    }
                }
                store_row<NV>(si.x_out + (long long)row * cols, cols, vec, lane, r);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
        const float mu = wave_sum(s) / cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = (i * 64 + lane) * 4 + t;
                const float d = c < cols ? el(r.v[i], t) - mu : 0.f;
                q += d * d;
            }
        const float rs = rsqrtf(wave_sum(q) / cols + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c0 = (i * 64 + lane) * 4;
            uint4 rnd = make_uint4(0, 0, 0, 0);
            if (p > 0.f) rnd = philox4x32_10(offset + (uint64_t)row * (NV * 64) + (i * 64 + lane), seed);
            uint32_t keep4 = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float o = (el(r.v[i], t) - mu) * rs * el(wv.v[i], t) + el(bv.v[i], t);
                if (relu) o = fmaxf(o, 0.f);
                if (p > 0.f) {
                    const uint32_t rr = t == 0 ? rnd.x : t == 1 ? rnd.y : t == 2 ? rnd.z : rnd.w;
                    const bool keep = u01(rr) >= p;
                    o = keep ? o * inv_keep : 0.f;
                    keep4 |= (keep ? 1u : 0u) << (8 * t);
                }
                el(r.v[i], t) = o;
            }
            if (p > 0.f && c0 < cols) {
                uint8_t* mp = mask + (long long)row * cols + c0;
                if (vec) *reinterpret_cast<uint32_t*>(mp) = keep4;
                else
                    for (int t = 0; t < 4 && c0 + t < cols; ++t) mp[t] = (keep4 >> (8 * t)) & 1;
            }
        }
        store_row<NV>(y + (long long)row * cols, cols, vec, lane, r);
        if (tee.lo) {
#pragma unroll
            for (int i = 0; i < NV; ++i) tee4(tee, row, (i * 64 + lane) * 4, cols, vec, r.v[i]);
        }
    }
}

template <int NV, typename T, bool FULL>
__global__ __launch_bounds__(256) void rowln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, T* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                        uint8_t* __restrict__ mask, int rows, int cols, float eps, int relu,
                                                        float p, uint64_t seed, uint64_t offset,
                                                        const uint64_t* __restrict__ dev_offset, const SplitTee tee,
                                                        const SlabInput si) {
    rowln_fwd_body<NV, T, FULL>(x, w, b, y, mean, rstd, mask, 0, rows, cols, eps, relu, p, seed, offset, dev_offset, blockIdx.x,
                                gridDim.x, tee, si);
}

// Grouped row LayerNorm: up to LN_MAX_GROUPS consecutive row ranges of ONE [rows, cols] matrix, each with its own affine
// parameters (the projection heads of the task batches), in one launch: blockIdx.y = range.  No dropout.
constexpr int LN_MAX_GROUPS = 4;
struct RowLNGroups {
    int row0[LN_MAX_GROUPS + 1];
    const float* w[LN_MAX_GROUPS];
    const float* b[LN_MAX_GROUPS];
};
template <int NV, typename T, bool FULL>
__global__ __launch_bounds__(256) void rowln_fwd_group_kernel(const T* __restrict__ x, const RowLNGroups G, T* __restrict__ y,
                                                              float* __restrict__ mean, float* __restrict__ rstd, int cols,
                                                              float eps, int relu, const SplitTee tee) {
    const int g = blockIdx.y;
    rowln_fwd_body<NV, T, FULL>(x, G.w[g], G.b[g], y, mean, rstd, nullptr, G.row0[g], G.row0[g + 1], cols, eps, relu, 0.f, 0, 0,
                                nullptr, blockIdx.x, gridDim.x, tee);
}

// dx for one row + per-wave column partials of dw/db, combined per workgroup through LDS.
template <int NV, typename T, bool FULL>
__device__ __forceinline__ void rowln_bwd_body(const T* __restrict__ dy, const T* __restrict__ x,
                                               const float* __restrict__ w, const float* __restrict__ b,
                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                               const uint8_t* __restrict__ mask, T* __restrict__ dx,
                                               float* __restrict__ ws, int row_begin, int rows, int cols, int relu, float p,
                                               int blk, int nblk, long long ws_blk) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [WPB][2][NV*256]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (FULL) cols = NV * 256;  // exact-width rows: every bounds check below folds away
    const bool vec = FULL || (cols & 3) == 0;
    Row<NV> wv, bv, dwp, dbp;
    load_row<NV>(w, cols, vec, lane, wv);
    load_row<NV>(b, cols, vec, lane, bv);
#pragma unroll
    for (int i = 0; i < NV; ++i) dwp.v[i] = dbp.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    auto process = [&](int row, Row<NV>& g, Row<NV>& xr) {
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c0 = (i * 64 + lane) * 4;
            uint32_t keep4 = 0x01010101u;
            if (p > 0.f && c0 < cols) {
                const uint8_t* mp = mask + (long long)row * cols + c0;
                if (vec) keep4 = *reinterpret_cast<const uint32_t*>(mp);
                else {
                    keep4 = 0;
                    for (int t = 0; t < 4 && c0 + t < cols; ++t) keep4 |= (uint32_t)mp[t] << (8 * t);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool in = c0 + t < cols;
                const float xh = in ? (el(xr.v[i], t) - mu) * rs : 0.f;
                float gg = in ? el(g.v[i], t) : 0.f;
                if (p > 0.f) gg = ((keep4 >> (8 * t)) & 1) ? gg * inv_keep : 0.f;
                if (relu && !(xh * el(wv.v[i], t) + el(bv.v[i], t) > 0.f)) gg = 0.f;
                el(dwp.v[i], t) += gg * xh;
                el(dbp.v[i], t) += gg;
                const float dxh = gg * el(wv.v[i], t);
                s1 += dxh;
                s2 += dxh * xh;
                el(g.v[i], t) = dxh;
                el(xr.v[i], t) = xh;
            }
        }
        s1 = wave_sum(s1) / cols;
        s2 = wave_sum(s2) / cols;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) el(g.v[i], t) = rs * (el(g.v[i], t) - s1 - el(xr.v[i], t) * s2);
        store_row<NV>(dx + (long long)row * cols, cols, vec, lane, g);
    };
    // TWO rows in flight per wave (a wave walks rows / (4 * grid) of them; with one row's loads outstanding at a
    // time the walk is latency bound).  Wide rows (NV > 4) keep one row in flight: registers.
    const RowWalk rw = row_walk(blk, nblk, row_begin, rows, wave, WPB);
    const int stride_rows = rw.step;
    for (int row = rw.first; row < rw.end; row += (NV <= 4 ? 2 : 1) * stride_rows) {
        Row<NV> g, xr;
        load_row<NV>(dy + (long long)row * cols, cols, vec, lane, g);
        load_row<NV>(x + (long long)row * cols, cols, vec, lane, xr);
        if constexpr (NV <= 4) {
            const int row2 = row + stride_rows;
            Row<NV> g2, xr2;
            if (row2 < rw.end) {
                load_row<NV>(dy + (long long)row2 * cols, cols, vec, lane, g2);
                load_row<NV>(x + (long long)row2 * cols, cols, vec, lane, xr2);
            }
            process(row, g, xr);
            if (row2 < rw.end) process(row2, g2, xr2);
        } else {
            process(row, g, xr);
        }
    }
    // combine the 4 waves' column partials in wave order, write this workgroup's partial row
    const int stride = NV * 256;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        *reinterpret_cast<float4*>(red + (wave * 2 + 0) * stride + c) = dwp.v[i];
        *reinterpret_cast<float4*>(red + (wave * 2 + 1) * stride + c) = dbp.v[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        float a = 0.f, d = 0.f;
#pragma unroll
        for (int wv_ = 0; wv_ < WPB; ++wv_) {
            a += red[(wv_ * 2 + 0) * stride + c];
            d += red[(wv_ * 2 + 1) * stride + c];
        }
        ws[(ws_blk * 2 + 0) * cols + c] = a;
        ws[(ws_blk * 2 + 1) * cols + c] = d;
    }
}

template <int NV, typename T, bool FULL>
__global__ __launch_bounds__(256) void rowln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const uint8_t* __restrict__ mask, T* __restrict__ dx,
                                                        float* __restrict__ ws, int rows, int cols, int relu, float p) {
    rowln_bwd_body<NV, T, FULL>(dy, x, w, b, mean, rstd, mask, dx, ws, 0, rows, cols, relu, p, blockIdx.x, gridDim.x, blockIdx.x);
}

// grouped backward: partial rows of range g land in ws[(g * gridDim.x + blockIdx.x)][2][cols]
template <int NV, typename T, bool FULL>
__global__ __launch_bounds__(256) void rowln_bwd_group_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const RowLNGroups G, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, T* __restrict__ dx,
                                                              float* __restrict__ ws, int cols, int relu) {
    const int g = blockIdx.y;
    rowln_bwd_body<NV, T, FULL>(dy, x, G.w[g], G.b[g], mean, rstd, nullptr, dx, ws, G.row0[g], G.row0[g + 1], cols, relu, 0.f,
                                blockIdx.x, gridDim.x, (long long)g * gridDim.x + blockIdx.x);
}


// ---- wide rows (1024 < cols <= 4096, cols a multiple of 1024: the shipped temporal pooling's hidden width 4096) ------------------
// One wave per row holds a 4096-column row as 16 float4 per lane -- with the affine rows and, backward, the two column-partial rows
// that is 256-384 registers per lane: one wave per SIMD, spills, a 150 MB backward pass at 270-380 us (profiles/r05 Hp = 4096
// timeline: the two launches were 650 us of a 3.2 ms step).  Here a WORKGROUP takes a row: wave v holds columns
// [v cols / 4, (v + 1) cols / 4) as NVW float4 per lane (NVW = cols / 1024 <= 4: the register budget of the 1024-column kernels), the
// four waves' row sums meet through LDS (wave order: reproducible), two rows in flight per workgroup.  The column partials of
// dw / db need no combining across waves: every wave owns its columns.  Same dropout mask bits as the one-wave kernel (the Philox
// counter of a column group does not depend on who draws it).
template <int NVW, typename T>
__global__ __launch_bounds__(256) void rowln_fwd_wide_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, T* __restrict__ y,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             uint8_t* __restrict__ mask, int rows, float eps, int relu, float p,
                                                             uint64_t seed, uint64_t offset, const uint64_t* __restrict__ dev_offset,
                                                             const SplitTee tee) {
    __shared__ float part[4][WPB];  // [row in flight x {sum, squares}][wave]
    constexpr int cols = NVW * 1024, WCOLS = NVW * 256;
    if (dev_offset) offset += dev_offset[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cw = wave * WCOLS;  // this wave's first column
    Row<NVW> wv, bv;
    load_row<NVW>(w + cw, WCOLS, true, lane, wv);
    load_row<NVW>(b + cw, WCOLS, true, lane, bv);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    auto finish = [&](int row, Row<NVW>& r, float mu, float rs) {
        if (threadIdx.x == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int c0 = cw + (i * 64 + lane) * 4;
            uint4 rnd = make_uint4(0, 0, 0, 0);
            // (the one-wave kernel's counter: row * (16 * 64) + column group, 16 = its float4 per lane for every width above 1024)
            if (p > 0.f) rnd = philox4x32_10(offset + (uint64_t)row * (16 * 64) + (uint64_t)(c0 >> 2), seed);
            uint32_t keep4 = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float o = (el(r.v[i], t) - mu) * rs * el(wv.v[i], t) + el(bv.v[i], t);
                if (relu) o = fmaxf(o, 0.f);
                if (p > 0.f) {
                    const uint32_t rr = t == 0 ? rnd.x : t == 1 ? rnd.y : t == 2 ? rnd.z : rnd.w;
                    const bool keep = u01(rr) >= p;
                    o = keep ? o * inv_keep : 0.f;
                    keep4 |= (keep ? 1u : 0u) << (8 * t);
                }
                el(r.v[i], t) = o;
            }
            if (p > 0.f) *reinterpret_cast<uint32_t*>(mask + (long long)row * cols + c0) = keep4;
        }
        store_row<NVW>(y + (long long)row * cols + cw, WCOLS, true, lane, r);
        if (tee.lo) {
#pragma unroll
            for (int i = 0; i < NVW; ++i) tee4(tee, row, cw + (i * 64 + lane) * 4, cols, true, r.v[i]);
        }
    };
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, 0, 1);  // (a workgroup per row; XCD x owns a contiguous eighth of them)
    for (int row = rw.first; row < rw.end; row += 2 * rw.step) {  // (block-uniform: the barriers below are met by all four waves)
        const int row2 = row + rw.step;
        const bool two = row2 < rw.end;
        Row<NVW> r, r2;
        load_row<NVW>(x + (long long)row * cols + cw, WCOLS, true, lane, r);
        if (two) load_row<NVW>(x + (long long)row2 * cols + cw, WCOLS, true, lane, r2);
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
            if (two) s2 += (r2.v[i].x + r2.v[i].y) + (r2.v[i].z + r2.v[i].w);
        }
        s = wave_sum(s);
        s2 = wave_sum(s2);
        if (lane == 0) {
            part[0][wave] = s;
            part[1][wave] = s2;
        }
        __syncthreads();
        const float mu = (((part[0][0] + part[0][1]) + part[0][2]) + part[0][3]) / cols;
        const float mu2 = (((part[1][0] + part[1][1]) + part[1][2]) + part[1][3]) / cols;
        float q = 0.f, q2 = 0.f;
#pragma unroll
        for (int i = 0; i < NVW; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float d = el(r.v[i], t) - mu, d2 = el(r2.v[i], t) - mu2;
                q += d * d;
                if (two) q2 += d2 * d2;
            }
        q = wave_sum(q);
        q2 = wave_sum(q2);
        if (lane == 0) {
            part[2][wave] = q;
            part[3][wave] = q2;
        }
        __syncthreads();
        const float rs = rsqrtf((((part[2][0] + part[2][1]) + part[2][2]) + part[2][3]) / cols + eps);
        const float rs2 = rsqrtf((((part[3][0] + part[3][1]) + part[3][2]) + part[3][3]) / cols + eps);
        finish(row, r, mu, rs);
        if (two) finish(row2, r2, mu2, rs2);
        __syncthreads();  // (the next pair of rows rewrites ``part``)
    }
}

template <int NVW, typename T>
__global__ __launch_bounds__(256) void rowln_bwd_wide_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const uint8_t* __restrict__ mask, T* __restrict__ dx,
                                                             float* __restrict__ ws, int rows, int relu, float p) {
    __shared__ float part[4][WPB];  // [row in flight x {s1, s2}][wave]
    constexpr int cols = NVW * 1024, WCOLS = NVW * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cw = wave * WCOLS;
    Row<NVW> wv, bv, dwp, dbp;
    load_row<NVW>(w + cw, WCOLS, true, lane, wv);
    load_row<NVW>(b + cw, WCOLS, true, lane, bv);
#pragma unroll
    for (int i = 0; i < NVW; ++i) dwp.v[i] = dbp.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    // first half of a row: g <- dy * w (gated), xr <- x_hat, the column partials, this wave's (s1, s2)
    auto head = [&](int row, Row<NVW>& g, Row<NVW>& xr, float& s1, float& s2) {
        const float mu = mean[row], rs = rstd[row];
        s1 = s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int c0 = cw + (i * 64 + lane) * 4;
            uint32_t keep4 = 0x01010101u;
            if (p > 0.f) keep4 = *reinterpret_cast<const uint32_t*>(mask + (long long)row * cols + c0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float xh = (el(xr.v[i], t) - mu) * rs;
                float gg = el(g.v[i], t);
                if (p > 0.f) gg = ((keep4 >> (8 * t)) & 1) ? gg * inv_keep : 0.f;
                if (relu && !(xh * el(wv.v[i], t) + el(bv.v[i], t) > 0.f)) gg = 0.f;
                el(dwp.v[i], t) += gg * xh;
                el(dbp.v[i], t) += gg;
                const float dxh = gg * el(wv.v[i], t);
                s1 += dxh;
                s2 += dxh * xh;
                el(g.v[i], t) = dxh;
                el(xr.v[i], t) = xh;
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
    };
    auto tail = [&](int row, Row<NVW>& g, const Row<NVW>& xr, float s1, float s2) {
        const float rs = rstd[row];
#pragma unroll
        for (int i = 0; i < NVW; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) el(g.v[i], t) = rs * (el(g.v[i], t) - s1 - el(xr.v[i], t) * s2);
        store_row<NVW>(dx + (long long)row * cols + cw, WCOLS, true, lane, g);
    };
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, 0, 1);
    for (int row = rw.first; row < rw.end; row += 2 * rw.step) {
        const int row2 = row + rw.step;
        const bool two = row2 < rw.end;
        Row<NVW> g, xr, g2, xr2;
        load_row<NVW>(dy + (long long)row * cols + cw, WCOLS, true, lane, g);
        load_row<NVW>(x + (long long)row * cols + cw, WCOLS, true, lane, xr);
        if (two) {
            load_row<NVW>(dy + (long long)row2 * cols + cw, WCOLS, true, lane, g2);
            load_row<NVW>(x + (long long)row2 * cols + cw, WCOLS, true, lane, xr2);
        }
        float a1, a2, b1 = 0.f, b2 = 0.f;
        head(row, g, xr, a1, a2);
        if (two) head(row2, g2, xr2, b1, b2);
        if (lane == 0) {
            part[0][wave] = a1;
            part[1][wave] = a2;
            part[2][wave] = b1;
            part[3][wave] = b2;
        }
        __syncthreads();
        const float t1 = (((part[0][0] + part[0][1]) + part[0][2]) + part[0][3]) / cols;
        const float t2 = (((part[1][0] + part[1][1]) + part[1][2]) + part[1][3]) / cols;
        const float u1 = (((part[2][0] + part[2][1]) + part[2][2]) + part[2][3]) / cols;
        const float u2 = (((part[3][0] + part[3][1]) + part[3][2]) + part[3][3]) / cols;
        tail(row, g, xr, t1, t2);
        if (two) tail(row2, g2, xr2, u1, u2);
        __syncthreads();
    }
    // this workgroup's partial rows: every wave owns its columns
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
        const int c = cw + (i * 64 + lane) * 4;
        *reinterpret_cast<float4*>(ws + ((long long)blockIdx.x * 2 + 0) * cols + c) = dwp.v[i];
        *reinterpret_cast<float4*>(ws + ((long long)blockIdx.x * 2 + 1) * cols + c) = dbp.v[i];
    }
}

// out_a[c] += sum_b ws[b][0][c]; out_b[c] += sum_b ws[b][1][c]   (fixed order)
// The order (unchanged since round 1, so the bits are): sixteen row groups rg = 0 .. 15, group rg sums rows rg + 16 u + 128 j for
// u = 0 .. 7 inside j = 0, 1, ...; the sixteen group sums are then added in group order.
// Sixteen waves per workgroup, wave = row group, a strip of 32 columns: lanes 0-31 sum the dw partials of the strip, lanes 32-63 the
// db partials (two 128-byte pieces per row), 8 rows in flight per lane; 32 workgroups per 1024-column problem.  The reduction
// sits between the step's last weight gradient and Adam; 16-column strips on 4-wave workgroups (64-byte pieces) took 13-17 us
// stand-alone, this takes 8-10 -- but beside the Adam launch that saturates HBM at that point of the step it still takes 13.8.
constexpr int RED_COLS = 32;
__device__ __forceinline__ void partial_reduce2_body(const float* __restrict__ ws, float* __restrict__ oa, float* __restrict__ ob,
                                                     int nblk, int cols, int strip) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int half = lane >> 5;  // 0: the first partial row of a pair (dw), 1: the second (db)
    const int c = strip * RED_COLS + (lane & 31);
    float acc = 0.f;
    if (c < cols)
        for (int k = rg; k < nblk; k += 128) {  // 8 loads in flight, summed in k order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k + 16 * u < nblk ? ws[((long long)(k + 16 * u) * 2 + half) * cols + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    red[rg][lane] = acc;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        float* o = half ? ob : oa;
        if (o) o[c] += t;
    }
}

__global__ __launch_bounds__(1024) void partial_reduce2_kernel(const float* __restrict__ ws, float* __restrict__ oa,
                                                              float* __restrict__ ob, int nblk, int cols) {
    partial_reduce2_body(ws, oa, ob, nblk, cols, blockIdx.x);
}

// several reductions of that kind in ONE launch (blockIdx.y = problem): the dw / db reductions of the norm layers whose
// backward has run since the last weight-gradient flush (each is ~4 MB of partial rows: launch latency, not bytes)
constexpr int REDUCE_MAX = 8;
struct ReduceBatch {
    const float* ws[REDUCE_MAX];
    float* oa[REDUCE_MAX];
    float* ob[REDUCE_MAX];
    int nblk[REDUCE_MAX];
    int cols[REDUCE_MAX];
};
__global__ __launch_bounds__(1024) void partial_reduce2_multi_kernel(const ReduceBatch B) {
    const int p = blockIdx.y;
    if ((int)blockIdx.x * RED_COLS >= B.cols[p]) return;  // (block-uniform: the grid is sized for the widest problem)
    partial_reduce2_body(B.ws[p], B.oa[p], B.ob[p], B.nblk[p], B.cols[p], blockIdx.x);
}

// -------------------------------------------------------------------------------------------
// graph-mode LayerNorm + LeakyReLU (statistics over all elements of a row segment)
// -------------------------------------------------------------------------------------------
constexpr int MAXSEG = 16;

__device__ __forceinline__ int seg_of(const int* __restrict__ seg_ptr, int n_seg, int row) {
    int s = 0;
    while (s + 1 < n_seg && row >= seg_ptr[s + 1]) ++s;
    return s;
}

// pass 1: per-workgroup per-segment (sum, sumsq) in double -> ws[blk][seg][2]
template <int NV, typename T, bool FULL>
__global__ __launch_bounds__(256) void graphln_stats_kernel(const T* __restrict__ x, const int* __restrict__ seg_ptr,
                                                            int n_seg, int rows, int cols, double* __restrict__ ws,
                                                            const SlabInput si = SlabInput{nullptr, nullptr, nullptr}) {
    __shared__ double acc[WPB][MAXSEG][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (FULL) cols = NV * 256;  // exact-width rows: every bounds check below folds away
    const bool vec = FULL || (cols & 3) == 0;
    if (lane < n_seg) acc[wave][lane][0] = acc[wave][lane][1] = 0.0;
    auto process = [&](int row, const Row<NV>& r) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = el(r.v[i], t);  // out-of-range columns were loaded as 0
                s += v;
                q += v * v;
            }
        const double ds = wave_sum((double)s), dq = wave_sum((double)q);
        if (lane == 0) {
            const int sg = seg_of(seg_ptr, n_seg, row);
            acc[wave][sg][0] += ds;
            acc[wave][sg][1] += dq;
        }
    };
    // the input as two K slabs + bias (egk_slab_input_next): gemm_splitk_reduce's arithmetic; the reduced row is stored for the
    // normalising launch that follows (and every later reader)
    auto slab_row = [&](int row, Row<NV>& r) {
        if constexpr (sizeof(T) == 4) {
            if (si.x2) {
                Row<NV> t;
                load_row<NV>(si.x2 + (long long)row * cols, cols, vec, lane, t);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    r.v[i].x += t.v[i].x; r.v[i].y += t.v[i].y; r.v[i].z += t.v[i].z; r.v[i].w += t.v[i].w;
                }
                if (si.bias) {
                    load_row<NV>(si.bias, cols, vec, lane, t);
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        r.v[i].x += t.v[i].x; r.v[i].y += t.v[i].y; r.v[i].z += t.v[i].z; r.v[i].w += t.v[i].w;
                    }
                }
                store_row<NV>(si.x_out + (long long)row * cols, cols, vec, lane, r);
            }
        }
    };
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    const int stride_rows = rw.step;  // two rows in flight per wave (see rowln_bwd_kernel)
    for (int row = rw.first; row < rw.end; row += (NV <= 4 ? 2 : 1) * stride_rows) {
        Row<NV> r;
        load_row<NV>(x + (long long)row * cols, cols, vec, lane, r);
        if constexpr (NV <= 4) {
            const int row2 = row + stride_rows;
            Row<NV> r2;
            if (row2 < rw.end) load_row<NV>(x + (long long)row2 * cols, cols, vec, lane, r2);
            slab_row(row, r);
            process(row, r);
            if (row2 < rw.end) {
                slab_row(row2, r2);
                process(row2, r2);
            }
        } else {
            slab_row(row, r);
            process(row, r);
        }
    }
    __syncthreads();
    if (threadIdx.x < n_seg * 2) {
        const int sg = threadIdx.x >> 1, k = threadIdx.x & 1;
        double t = 0.0;
        for (int wv_ = 0; wv_ < WPB; ++wv_) t += acc[wv_][sg][k];
        ws[((long long)blockIdx.x * n_seg + sg) * 2 + k] = t;
    }
}

// Block-wide sums of the per-workgroup double partials ws[k][sg][j], k < nblk, for EVERY (sg, j) at once, in a fixed
// order (thread-strided partial sums, wave shuffle tree, 4 waves combined in wave order): bitwise reproducible.
// All loads of a group of 8 values are issued before the first reduction and the group shares ONE barrier pair: this
// prologue runs in every workgroup of the normalising kernels, and a chain of 2 * n_seg dependent
// load -> reduce -> barrier rounds used to cost more than the row work behind it.
__device__ __forceinline__ void block_sum_all(const double* __restrict__ ws, int nblk, int n_seg, double* out /* [n_seg*2] LDS */,
                                              double (*scratch)[8] /* [4][8] LDS */) {
    const int nv = n_seg * 2;
    for (int v0 = 0; v0 < nv; v0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = 0.0;
        for (int k = threadIdx.x; k < nblk; k += 256) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v0 + u < nv) v[u] += ws[(long long)k * nv + v0 + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = wave_sum(v[u]);
        __syncthreads();  // scratch free (previous group consumed)
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) scratch[threadIdx.x >> 6][u] = v[u];
        }
        __syncthreads();
        if (threadIdx.x < 8 && v0 + threadIdx.x < nv)
            out[v0 + threadIdx.x] = (scratch[0][threadIdx.x] + scratch[1][threadIdx.x]) + (scratch[2][threadIdx.x] + scratch[3][threadIdx.x]);
    }
    __syncthreads();
}

// every workgroup re-derives (mean, 1/(std+eps)) of every segment from the partials
__device__ __forceinline__ void graphln_finish_stats(const double* __restrict__ ws, int nblk, const int* __restrict__ seg_ptr,
                                                     int n_seg, int cols, float eps, float (*st)[2]) {
    __shared__ double scratch[4][8];
    __shared__ double sums[MAXSEG * 2];
    block_sum_all(ws, nblk, n_seg, sums, scratch);
    if (threadIdx.x < n_seg) {
        const int sg = threadIdx.x;
        const double s = sums[sg * 2 + 0], q = sums[sg * 2 + 1];
        const double n = (double)(seg_ptr[sg + 1] - seg_ptr[sg]) * cols;
        const double mu = n > 0 ? s / n : 0.0;
        double var = n > 0 ? q / n - mu * mu : 0.0;
        if (var < 0) var = 0;
        st[sg][0] = (float)mu;
        st[sg][1] = (float)(1.0 / (sqrt(var) + (double)eps));
    }
    __syncthreads();
}

template <int NV, typename T, bool FULL>
__global__ __launch_bounds__(256) void graphln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, T* __restrict__ y,
                                                          float* __restrict__ stats, const int* __restrict__ seg_ptr,
                                                          int n_seg, int rows, int cols, float eps, float slope,
                                                          const double* __restrict__ ws, int nblk_stats, const SplitTee tee) {
    __shared__ float st[MAXSEG][2];
    graphln_finish_stats(ws, nblk_stats, seg_ptr, n_seg, cols, eps, st);
    if (blockIdx.x == 0 && threadIdx.x < n_seg * 2) stats[threadIdx.x] = st[threadIdx.x >> 1][threadIdx.x & 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (FULL) cols = NV * 256;  // exact-width rows: every bounds check below folds away
    const bool vec = FULL || (cols & 3) == 0;
    Row<NV> wv, bv;
    load_row<NV>(w, cols, vec, lane, wv);
    load_row<NV>(b, cols, vec, lane, bv);
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const int sg = seg_of(seg_ptr, n_seg, row);
        const float mu = st[sg][0], ri = st[sg][1];
        Row<NV> r;
        load_row<NV>(x + (long long)row * cols, cols, vec, lane, r);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float o = (el(r.v[i], t) - mu) * ri * el(wv.v[i], t) + el(bv.v[i], t);
                el(r.v[i], t) = o > 0.f ? o : o * slope;
            }
        store_row<NV>(y + (long long)row * cols, cols, vec, lane, r);
        if (tee.lo) {
#pragma unroll
            for (int i = 0; i < NV; ++i) tee4(tee, row, (i * 64 + lane) * 4, cols, vec, r.v[i]);
        }
    }
}

// bwd pass 1: per-segment S1 = sum(dxhat), S2 = sum(dxhat*xhat) (double) + column partials of dw/db
template <int NV, typename T, bool FULL>
__global__ __launch_bounds__(256) void graphln_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                const float* __restrict__ stats,
                                                                const int* __restrict__ seg_ptr, int n_seg, int rows,
                                                                int cols, float slope, double* __restrict__ ws_seg,
                                                                float* __restrict__ ws_col) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [WPB][2][NV*256]
    __shared__ double acc[WPB][MAXSEG][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (FULL) cols = NV * 256;  // exact-width rows: every bounds check below folds away
    const bool vec = FULL || (cols & 3) == 0;
    if (lane < n_seg) acc[wave][lane][0] = acc[wave][lane][1] = 0.0;
    Row<NV> wv, bv, dwp, dbp;
    load_row<NV>(w, cols, vec, lane, wv);
    load_row<NV>(b, cols, vec, lane, bv);
#pragma unroll
    for (int i = 0; i < NV; ++i) dwp.v[i] = dbp.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto process = [&](int row, const Row<NV>& g, const Row<NV>& xr) {
        const int sg = seg_of(seg_ptr, n_seg, row);
        const float mu = stats[sg * 2 + 0], ri = stats[sg * 2 + 1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool in = (i * 64 + lane) * 4 + t < cols;
                const float xh = in ? (el(xr.v[i], t) - mu) * ri : 0.f;
                const float pre = xh * el(wv.v[i], t) + el(bv.v[i], t);
                const float gg = in ? el(g.v[i], t) * (pre > 0.f ? 1.f : slope) : 0.f;
                el(dwp.v[i], t) += gg * xh;
                el(dbp.v[i], t) += gg;
                const float dxh = gg * el(wv.v[i], t);
                s1 += dxh;
                s2 += dxh * xh;
            }
        const double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
        if (lane == 0) {
            acc[wave][sg][0] += d1;
            acc[wave][sg][1] += d2;
        }
    };
    // two rows in flight per wave (see rowln_bwd_kernel)
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);
    const int stride_rows = rw.step;
    for (int row = rw.first; row < rw.end; row += (NV <= 4 ? 2 : 1) * stride_rows) {
        Row<NV> g, xr;
        load_row<NV>(dy + (long long)row * cols, cols, vec, lane, g);
        load_row<NV>(x + (long long)row * cols, cols, vec, lane, xr);
        if constexpr (NV <= 4) {
            const int row2 = row + stride_rows;
            Row<NV> g2, xr2;
            if (row2 < rw.end) {
                load_row<NV>(dy + (long long)row2 * cols, cols, vec, lane, g2);
                load_row<NV>(x + (long long)row2 * cols, cols, vec, lane, xr2);
            }
            process(row, g, xr);
            if (row2 < rw.end) process(row2, g2, xr2);
        } else {
            process(row, g, xr);
        }
    }
    const int stride = NV * 256;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        *reinterpret_cast<float4*>(red + (wave * 2 + 0) * stride + c) = dwp.v[i];
        *reinterpret_cast<float4*>(red + (wave * 2 + 1) * stride + c) = dbp.v[i];
    }
    __syncthreads();
    if (threadIdx.x < n_seg * 2) {
        const int sg = threadIdx.x >> 1, k = threadIdx.x & 1;
        double t = 0.0;
        for (int wv_ = 0; wv_ < WPB; ++wv_) t += acc[wv_][sg][k];
        ws_seg[((long long)blockIdx.x * n_seg + sg) * 2 + k] = t;
    }
    for (int c = threadIdx.x; c < cols; c += 256) {
        float a = 0.f, d = 0.f;
#pragma unroll
        for (int wv_ = 0; wv_ < WPB; ++wv_) {
            a += red[(wv_ * 2 + 0) * stride + c];
            d += red[(wv_ * 2 + 1) * stride + c];
        }
        ws_col[((long long)blockIdx.x * 2 + 0) * cols + c] = a;
        ws_col[((long long)blockIdx.x * 2 + 1) * cols + c] = d;
    }
}

// bwd pass 2: dx = r*dxhat - r*S1/n - xhat*S2/(n*sigma),  sigma = 1/r - eps
// COLS: also the per-workgroup column partials of dw / db (ws_col[blk][2][cols]) -- when the segment sums came from the
// epilogue of the contraction that produced dy (egk_gemm_desc.st_mode 2) there is no statistics pass to carry them
template <int NV, typename T, bool FULL, bool COLS = false>
__global__ __launch_bounds__(256) void graphln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          const float* __restrict__ stats, T* __restrict__ dx,
                                                          const int* __restrict__ seg_ptr, int n_seg, int rows, int cols,
                                                          float eps, float slope, const double* __restrict__ ws_seg,
                                                          int nblk_stats, float* __restrict__ ws_col = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // COLS: [WPB][2][NV*256]
    __shared__ float sc[MAXSEG][4];  // mean, r, r*S1/n, S2/(n*sigma)
    __shared__ double scratch[4][8];
    __shared__ double sums[MAXSEG * 2];
    block_sum_all(ws_seg, nblk_stats, n_seg, sums, scratch);
    if (threadIdx.x < n_seg) {
        const int sg = threadIdx.x;
        const double s1 = sums[sg * 2 + 0], s2 = sums[sg * 2 + 1];
        const double n = (double)(seg_ptr[sg + 1] - seg_ptr[sg]) * cols;
        const double r = stats[sg * 2 + 1];
        const double sigma = 1.0 / r - (double)eps;
        sc[sg][0] = stats[sg * 2 + 0];
        sc[sg][1] = (float)r;
        sc[sg][2] = n > 0 ? (float)(r * s1 / n) : 0.f;
        sc[sg][3] = (n > 0 && sigma > 0) ? (float)(s2 / (n * sigma)) : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (FULL) cols = NV * 256;  // exact-width rows: every bounds check below folds away
    const bool vec = FULL || (cols & 3) == 0;
    Row<NV> wv, bv, dwp, dbp;
    load_row<NV>(w, cols, vec, lane, wv);
    load_row<NV>(b, cols, vec, lane, bv);
    if constexpr (COLS) {
#pragma unroll
        for (int i = 0; i < NV; ++i) dwp.v[i] = dbp.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const int sg = seg_of(seg_ptr, n_seg, row);
        const float mu = sc[sg][0], ri = sc[sg][1], c1 = sc[sg][2], c2 = sc[sg][3];
        Row<NV> g, xr;
        load_row<NV>(dy + (long long)row * cols, cols, vec, lane, g);
        load_row<NV>(x + (long long)row * cols, cols, vec, lane, xr);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool in = FULL || (i * 64 + lane) * 4 + t < cols;
                const float xh = in ? (el(xr.v[i], t) - mu) * ri : 0.f;
                const float pre = xh * el(wv.v[i], t) + el(bv.v[i], t);
                const float gg = in ? el(g.v[i], t) * (pre > 0.f ? 1.f : slope) : 0.f;
                if constexpr (COLS) {
                    el(dwp.v[i], t) += gg * xh;
                    el(dbp.v[i], t) += gg;
                }
                const float dxh = gg * el(wv.v[i], t);
                el(g.v[i], t) = ri * dxh - c1 - xh * c2;
            }
        store_row<NV>(dx + (long long)row * cols, cols, vec, lane, g);
    }
    if constexpr (COLS) {  // the 4 waves' column partials in wave order -> this workgroup's partial row (as in the stats pass)
        const int stride = NV * 256;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            *reinterpret_cast<float4*>(red + (wave * 2 + 0) * stride + c) = dwp.v[i];
            *reinterpret_cast<float4*>(red + (wave * 2 + 1) * stride + c) = dbp.v[i];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256) {
            float a = 0.f, d = 0.f;
#pragma unroll
            for (int wv_ = 0; wv_ < WPB; ++wv_) {
                a += red[(wv_ * 2 + 0) * stride + c];
                d += red[(wv_ * 2 + 1) * stride + c];
            }
            ws_col[((long long)blockIdx.x * 2 + 0) * cols + c] = a;
            ws_col[((long long)blockIdx.x * 2 + 1) * cols + c] = d;
        }
    }
}

// -------------------------------------------------------------------------------------------
// column sum: stage 1 partials over row chunks, stage 2 fixed-order sum
// -------------------------------------------------------------------------------------------
// stage 1: each lane owns 4 consecutive columns (8 / 16 B loads), each workgroup a chunk of rows
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long long ldx, int M, int N,
                                                             float* __restrict__ ws, int rows_per, int vec) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= N) return;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    // 8 row loads in flight per lane (a lone dependent load per iteration is latency bound: ~0.5 us per row)
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0; r < r1; r += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = r + u < r1 ? ld4t(x + (long long)(r + u) * ldx, c, N, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 3].x += v[u].x; acc[u & 3].y += v[u].y; acc[u & 3].z += v[u].z; acc[u & 3].w += v[u].w;
        }
    }
    st4t(ws + (long long)blockIdx.y * N, c, N, (N & 3) == 0,
         make_float4((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y),
                     (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)));
}
// stage 2: 32 columns x 8 partial-row groups per workgroup, groups combined in LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int N,
                                                           int nchunk, int accumulate) {
    __shared__ float red[8][32];
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cg;
    float s = 0.f;
    if (c < N)
        for (int k = rg; k < nchunk; k += 64) {  // 8 loads in flight, summed in k order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k + 8 * u < nchunk ? ws[(long long)(k + 8 * u) * N + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    red[rg][cg] = s;
    __syncthreads();
    if (rg == 0 && c < N) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][cg];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// -------------------------------------------------------------------------------------------
// One-logit classifier + BCE-with-logits, forward AND backward in one pass over the rows (the PNR head: reference
// models/tasks/pnr.py:20,37-52 -- Linear(features, 1) -> squeeze -> BCEWithLogitsLoss(reduction='none') at
// main_temporal.py:117-121).  A [rows, cols] x [cols, 1] contraction is a row reduction, not matrix work:
//   z_n = <f_n, w> + b;   loss_n = (1 - y_n) z_n + max(-z_n, 0) + log1p(exp(-|z_n|))      (egk_bce_fwd's formula)
//   g_n = (sigmoid(z_n) - y_n) * seed   (egk_bce_bwd's; rounded to T, as the contraction path rounds its operand)
//   df_n = g_n * w;   dw = sum_n g_n f_n;   db = sum_n g_n
// dw / db leave as per-workgroup partial rows ws[blk][cols + 4] (column ``cols`` = db), summed in block order by
// rowdot_reduce_kernel: fixed order, no atomics.
template <int NV, typename T, bool FULL>
__global__ __launch_bounds__(256) void rowdot_bce_kernel(const T* __restrict__ f, const T* __restrict__ w, const float* __restrict__ bias,
                                                         const long long* __restrict__ y, float* __restrict__ logits,
                                                         float* __restrict__ loss, T* __restrict__ df, float* __restrict__ ws,
                                                         int rows, int cols, float seed) {
    extern __shared__ __attribute__((aligned(16))) float red_[];  // [WPB][NV * 256 + 4]
    float (*red)[NV * 256 + 4] = reinterpret_cast<float (*)[NV * 256 + 4]>(red_);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (FULL) cols = NV * 256;
    const bool vec = FULL || (cols & 3) == 0;
    Row<NV> wv, acc;
    load_row<NV>(w, cols, vec, lane, wv);
#pragma unroll
    for (int i = 0; i < NV; ++i) acc.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float gsum = 0.f;
    const float b0 = bias ? bias[0] : 0.f;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);
    for (int row = rw.first; row < rw.end; row += rw.step) {
        Row<NV> r;
        load_row<NV>(f + (long long)row * cols, cols, vec, lane, r);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) d = fmaf(el(r.v[i], t), el(wv.v[i], t), d);  // out-of-range columns load as 0
        const float z = wave_sum(d) + b0;
        const float t = (float)y[row];
        if (lane == 0) {
            logits[row] = z;
            loss[row] = (1.f - t) * z + fmaxf(-z, 0.f) + log1pf(expf(-fabsf(z)));
        }
        if (df) {
            T gr;
            st1t(&gr, (1.f / (1.f + expf(-z)) - t) * seed);  // the gradient in the operand element type
            const float g = ld1t(&gr);
            gsum += g;  // (identical in every lane)
            Row<NV> o;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    el(o.v[i], t4) = g * el(wv.v[i], t4);
                    el(acc.v[i], t4) = fmaf(g, el(r.v[i], t4), el(acc.v[i], t4));
                }
            store_row<NV>(df + (long long)row * cols, cols, vec, lane, o);
        }
    }
    if (!df) return;
    // per-workgroup partial row: the four waves' accumulators summed in wave order
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) red[wave][(i * 64 + lane) * 4 + t] = el(acc.v[i], t);
    if (lane == 0) red[wave][NV * 256] = gsum;
    __syncthreads();
    float* out = ws + (long long)blockIdx.x * (cols + 4);
    for (int c = threadIdx.x; c <= cols; c += 256) {
        const int cc = c < cols ? c : NV * 256;
        out[c] = (red[0][cc] + red[1][cc]) + (red[2][cc] + red[3][cc]);
    }
}

// dw[c] += sum over blocks of ws[blk][c] (c < cols);  db[0] += sum of ws[blk][cols]
// workgroup = 16 columns x 16 block groups, 8 loads in flight per thread, the 16 partial sums of a column combined in LDS
// in group order (the shape of partial_reduce2_kernel: a single thread walking 512 blocks is a 125-us chain of loads)
__global__ __launch_bounds__(256) void rowdot_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db,
                                                            int nblk, int cols) {
    __shared__ float red[16][16];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cg;
    const long long ld = cols + 4;
    float a = 0.f;
    if (c <= cols)
        for (int k = rg; k < nblk; k += 128) {
            float va[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) va[u] = k + 16 * u < nblk ? ws[(long long)(k + 16 * u) * ld + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += va[u];
        }
    red[rg][cg] = a;
    __syncthreads();
    if (rg == 0 && c <= cols) {
        a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][cg];
        if (c < cols) dw[c] += a;
        else if (db) db[0] += a;
    }
}

static inline int nv_for(int cols) { return cols <= 256 ? 1 : cols <= 1024 ? 4 : cols <= 4096 ? 16 : 0; }
// development knobs (egk_tune 1 / 2).  Streaming kernels: three workgroups per CU -- every wave re-reads the f32 affine rows
// (8 KB against a 2 KB bf16 row) and, in the graph LayerNorm, re-reduces the statistics partials, so a wave should walk
// >= 2 rows; in-step A/B of the headline workload, 200 steps x 3 rounds: 2048 -> 1.637, 512 -> 1.626, 768 -> 1.612 ms
static int g_cap_partial = 512, g_cap_wide = 768;
extern int g_graph_rows_v2;  // (graph_ops.hip)
static inline int row_grid(int rows) {  // kernels that emit per-workgroup partial rows: two workgroups per CU
    int g = cdiv(rows, WPB);
    return g < 1 ? 1 : (g > g_cap_partial ? g_cap_partial : g);
}
static inline int row_grid_wide(int rows) {  // pure streaming row kernels
    int g = cdiv(rows, WPB);
    return g < 1 ? 1 : (g > g_cap_wide ? g_cap_wide : g);
}

// ---- two-logit classifier + cross entropy over FEW rows (the OSCC head: one row per sequence), loss AND gradients in one launch --
// logits[r, c] = f[r, :] . w[c, :] + bias[c];  loss[r] = lse - (1 - eps) z_y - eps / 2 (z_0 + z_1)  (0 for y = -1): ce_fwd_kernel;
// g[r, c] = seed (softmax_c - (c == y ? 1 - eps : 0) - eps / 2) rounded to T (ce_bwd_kernel + the operand cast of the contraction
// path);  df[r, :] = g[r, 0] w[0, :] + g[r, 1] w[1, :];  dw[c, :] += sum_r g[r, c] f[r, :];  db[c] += sum_r g[r, c], rows in order.
// TWO launches (a workgroup per row; a workgroup per 256 columns and source) -- eleven launches of the
// contraction path (a split 16 x 2 x 1024 contraction and its reduce, loss, loss gradient, cast, dX, two column sums, dW) sat on
// the critical path of BASELINE config 5 between the backbone's forward and backward.
constexpr int CE2_MAX_ROWS = 256, CE2_MAX_SRC = 4;
// S sources (the EgoPack head: the primary pooled features and one pooled GraphONE feature per auxiliary task, each with its own
// classifier; models/tasks/oscc.py:70-78): z = scale * sum_k (f_k w_k^T + bias_k), scale = 1 / S for ``average_logits``; every
// source receives the SAME rounded logit gradient g (the contraction path casts scale * dlogits once per source: equal values).
struct CE2Sources {
    const void* f[CE2_MAX_SRC];
    const void* w[CE2_MAX_SRC];
    const float* bias[CE2_MAX_SRC];
    void* df[CE2_MAX_SRC];
    float* dw[CE2_MAX_SRC];
    float* db[CE2_MAX_SRC];
    int n;
};
// Launch 1, one workgroup per ROW, wave k = source k: the row's dot products (a wave walks its source's row, 4 columns per lane and
// step), the fused logits, the loss, the rounded logit gradient g (also kept in ``gws`` [rows][2] for launch 2), and df_k = g w_k.
template <typename T>
__global__ __launch_bounds__(256) void rowdot_ce2_rows_kernel(const CE2Sources S, const long long* __restrict__ y,
                                                              float* __restrict__ logits, float* __restrict__ loss,
                                                              float* __restrict__ gws, int cols, float scale, float smoothing,
                                                              float seed) {
    __shared__ float part[CE2_MAX_SRC][2];
    __shared__ float gsh[2];
    const int row = blockIdx.x, lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    const bool vec = (cols & 3) == 0;
    const bool on = k < S.n;
    const T* __restrict__ f = on ? (const T*)S.f[k] + (long long)row * cols : nullptr;
    const T* __restrict__ w = on ? (const T*)S.w[k] : nullptr;
    float p0 = 0.f, p1 = 0.f;
    if (on)
        for (int c = lane * 4; c < cols; c += 256) {
            const float4 x = ld4t(f, c, cols, vec), w0 = ld4t(w, c, cols, vec), w1 = ld4t(w + cols, c, cols, vec);
            p0 = fmaf(x.x, w0.x, p0); p0 = fmaf(x.y, w0.y, p0); p0 = fmaf(x.z, w0.z, p0); p0 = fmaf(x.w, w0.w, p0);
            p1 = fmaf(x.x, w1.x, p1); p1 = fmaf(x.y, w1.y, p1); p1 = fmaf(x.z, w1.z, p1); p1 = fmaf(x.w, w1.w, p1);
        }
    p0 = wave_sum(p0);
    p1 = wave_sum(p1);
    if (lane == 0) {
        part[k][0] = on ? p0 + (S.bias[k] ? S.bias[k][0] : 0.f) : 0.f;
        part[k][1] = on ? p1 + (S.bias[k] ? S.bias[k][1] : 0.f) : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s0 = 0.f, s1 = 0.f;
        for (int q = 0; q < S.n; ++q) {  // source order
            s0 += part[q][0];
            s1 += part[q][1];
        }
        const float z0 = scale * s0, z1 = scale * s1;
        logits[row * 2 + 0] = z0;
        logits[row * 2 + 1] = z1;
        const float sm = smoothing > 0.f ? smoothing * 0.5f : 0.f;
        const float mx = fmaxf(z0, z1);
        const float l = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        const long long t = y[row];
        const bool live = t >= 0 && t < 2;
        loss[row] = live ? l - (1.f - smoothing) * (t == 0 ? z0 : z1) - (smoothing > 0.f ? sm * (z0 + z1) : 0.f) : 0.f;
        T g0, g1;  // the gradient of every source's logits, in the operand element type
        st1t(&g0, live ? scale * (seed * (expf(z0 - l) - (t == 0 ? 1.f - smoothing : 0.f) - sm)) : 0.f);
        st1t(&g1, live ? scale * (seed * (expf(z1 - l) - (t == 1 ? 1.f - smoothing : 0.f) - sm)) : 0.f);
        gsh[0] = ld1t(&g0);
        gsh[1] = ld1t(&g1);
        if (gws) {
            gws[row * 2 + 0] = gsh[0];
            gws[row * 2 + 1] = gsh[1];
        }
    }
    if (!gws) return;  // (uniform: forward only)
    __syncthreads();
    if (on && S.df[k]) {
        const float g0 = gsh[0], g1 = gsh[1];
        T* __restrict__ df = (T*)S.df[k] + (long long)row * cols;
        for (int c = lane * 4; c < cols; c += 256) {
            const float4 w0 = ld4t(w, c, cols, vec), w1 = ld4t(w + cols, c, cols, vec);
            st4t(df, c, cols, vec,
                 make_float4(fmaf(g1, w1.x, g0 * w0.x), fmaf(g1, w1.y, g0 * w0.y), fmaf(g1, w1.z, g0 * w0.z), fmaf(g1, w1.w, g0 * w0.w)));
        }
    }
}

// Launch 2, workgroup = (256 columns, source): dw_k[c, :] += sum_r g[r, c] f_k[r, :] and db_k += sum_r g[r, :], rows in order,
// sixteen rows in flight per lane.
template <typename T>
__global__ __launch_bounds__(64) void rowdot_ce2_cols_kernel(const CE2Sources S, const float* __restrict__ gws, int rows, int cols) {
    constexpr int RC = 16;
    const int k = blockIdx.y, c = (blockIdx.x * 64 + threadIdx.x) * 4;
    const bool vec = (cols & 3) == 0;
    const T* __restrict__ f = (const T*)S.f[k];
    float* __restrict__ dw = S.dw[k];
    if (dw && c < cols) {
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        for (int r0 = 0; r0 < rows; r0 += RC) {
            float4 x[RC];
#pragma unroll
            for (int r = 0; r < RC; ++r)
                x[r] = r0 + r < rows ? ld4t(f + (long long)(r0 + r) * cols, c, cols, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < RC; ++r) {
                if (r0 + r >= rows) break;
                const float g0 = gws[(r0 + r) * 2 + 0], g1 = gws[(r0 + r) * 2 + 1];
                a0.x = fmaf(g0, x[r].x, a0.x); a0.y = fmaf(g0, x[r].y, a0.y); a0.z = fmaf(g0, x[r].z, a0.z); a0.w = fmaf(g0, x[r].w, a0.w);
                a1.x = fmaf(g1, x[r].x, a1.x); a1.y = fmaf(g1, x[r].y, a1.y); a1.z = fmaf(g1, x[r].z, a1.z); a1.w = fmaf(g1, x[r].w, a1.w);
            }
        }
        const float v0[4] = {a0.x, a0.y, a0.z, a0.w}, v1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (c + t < cols) {
                dw[c + t] += v0[t];
                dw[cols + c + t] += v1[t];
            }
    }
    if (S.db[k] && blockIdx.x == 0 && threadIdx.x < 2) {
        float t = 0.f;
        for (int row = 0; row < rows; ++row) t += gws[row * 2 + threadIdx.x];
        S.db[k][threadIdx.x] += t;
    }
}

}  // namespace egk

using namespace egk;

#define DISPATCH_NV_(cols, ...)                                         \
    switch (nv_for(cols)) {                                             \
        case 1: { constexpr int NV = 1; constexpr bool FULL = false; __VA_ARGS__; } break;           \
        case 4:                                                                                    \
            if ((cols) == 1024) { constexpr int NV = 4; constexpr bool FULL = true; __VA_ARGS__; }   \
            else { constexpr int NV = 4; constexpr bool FULL = false; __VA_ARGS__; }                 \
            break;                                                                                 \
        case 16: { constexpr int NV = 16; constexpr bool FULL = false; __VA_ARGS__; } break;         \
        default: set_error("row width %d > 4096 unsupported", cols); return EGK_EUNSUPPORTED; \
    }
// NV (registers per lane) x T (activation element type)
#define DISPATCH_NV(cols, dtype, ...) EGK_DISPATCH_T(dtype, DISPATCH_NV_(cols, __VA_ARGS__))
// the workgroup-per-row kernels: cols = 2048 / 3072 / 4096
#define DISPATCH_NVW_(cols, ...)                                   \
    switch ((cols) / 1024) {                                       \
        case 2: { constexpr int NVW = 2; __VA_ARGS__; } break;     \
        case 3: { constexpr int NVW = 3; __VA_ARGS__; } break;     \
        default: { constexpr int NVW = 4; __VA_ARGS__; } break;    \
    }
static int g_wide_rows = 1;  // development knob (egk_tune 7): 0 = the one-wave-per-row kernels for every width
static inline bool wide_rows_ok(int cols, const void* in, const void* out, const float* w, const float* b, const uint8_t* mask, int dtype) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    (void)dtype;
    return g_wide_rows && cols > 1024 && cols <= 4096 && cols % 1024 == 0 && al(in) && al(out) && al(w) && al(b) &&
           (!mask || (reinterpret_cast<uintptr_t>(mask) & 3) == 0);
}

extern "C" {

// development knob: 1 = workgroup cap of the row kernels that emit per-workgroup partials, 2 = cap of the streaming ones
int egk_tune(int32_t key, int32_t value) {
    if (key == 1) { const int p = g_cap_partial; g_cap_partial = value; return p; }
    if (key == 2) { const int p = g_cap_wide; g_cap_wide = value; return p; }
    if (key == 3) { const int p = g_graph_rows_v2; g_graph_rows_v2 = value; return p; }  // graph_ops.hip: the rows1024.h kernels
    if (key == 5) { ::egk::set_zero_fill_blocks(value); return 0; }  // loss_optim.hip: workgroups of egk_zero_fill (0 = default)
    if (key == 6) { ::egk::set_adam_blocks(value); return 0; }       // loss_optim.hip: workgroup cap of the Adam launch (0 = default)
    if (key == 7) { const int p = g_wide_rows; g_wide_rows = value; return p; }  // the workgroup-per-row LayerNorm kernels on / off
    return -1;
}

static inline int colsum_chunks(int M, int N) {
    // enough workgroups to stream at HBM rate (>= ~256 with the column blocks), <= 64 partial rows to re-read
    const int col_blocks = cdiv(N, 1024);
    int chunks = cdiv(256, col_blocks);  // one workgroup per CU, 8 rows in flight per lane; 256 partial rows to re-read
    if (chunks > cdiv(M, 8)) chunks = cdiv(M, 8);
    return chunks < 1 ? 1 : chunks;
}

int egk_colsum_ws_len(int32_t M, int32_t N) { return colsum_chunks(M, N) * N; }

int egk_colsum(egk_stream_t stream, const void* x, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate,
               float* ws, int32_t dtype) {
    EGK_REQUIRE(x && out && ws, "egk_colsum: null pointer");
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int chunks = colsum_chunks(M, N);
    const int rows_per = cdiv(M, chunks);
    const int ebytes = dtype == EGK_BF16 ? 2 : 4;
    const int vec = (((uintptr_t)x) % (4 * ebytes) == 0) && ((ldx * ebytes) % (4 * ebytes) == 0);
    ProfScope prof(KID_COLSUM, s, 0, (dtype == EGK_BF16 ? 2.0 : 4.0) * M * N);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_partial_kernel<T>, dim3(cdiv(N, 1024), chunks), dim3(256), 0, s,
                                             (const T*)x, (long long)ldx, M, N, ws, rows_per, vec));
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 32)), dim3(256), 0, s, ws, out, N, chunks, accumulate);
    return check_launch("egk_colsum");
}

int egk_rowln_fwd(egk_stream_t stream, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                  uint8_t* mask, int32_t rows, int32_t cols, float eps, int32_t relu, float p, uint64_t seed,
                  uint64_t offset, const uint64_t* dev_offset, int32_t dtype) {
    EGK_REQUIRE(x && w && b && y && mean && rstd, "egk_rowln_fwd: null pointer");
    EGK_REQUIRE(p == 0.f || mask, "egk_rowln_fwd: dropout needs a mask buffer");
    EGK_REQUIRE(p >= 0.f && p < 1.f, "egk_rowln_fwd: p out of range");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    ProfScope prof(KID_ROWLN_FWD, s, 0, 2 * eb * rows * cols + (p > 0 ? 1.0 * rows * cols : 0));
    const SplitTee tee = take_split_tee();
    EGK_REQUIRE(!tee.lo || dtype == EGK_F32, "egk_rowln_fwd: a split tee needs an f32 result");
    const SlabInput si = take_slab_input();
    EGK_REQUIRE(!si.x2 || (dtype == EGK_F32 && cols % 4 == 0 && cols <= 1024), "egk_rowln_fwd: a slab input needs f32 rows of <= 1024 columns");
    if (!si.x2 && wide_rows_ok(cols, x, y, w, b, mask, dtype)) {
        const int grid = rows < 1536 ? rows : 1536;  // (six workgroups per CU: two rows in flight each)
        EGK_DISPATCH_T(dtype, DISPATCH_NVW_(cols, hipLaunchKernelGGL((rowln_fwd_wide_kernel<NVW, T>), dim3(grid), dim3(256), 0, s, (const T*)x, w,
                                                                    b, (T*)y, mean, rstd, mask, rows, eps, relu, p, seed, offset,
                                                                    dev_offset, tee)));
        return check_launch("egk_rowln_fwd");
    }
    DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((rowln_fwd_kernel<NV, T, FULL>), dim3(row_grid_wide(rows)), dim3(256), 0, s, (const T*)x, w, b,
                                                 (T*)y, mean, rstd, mask, rows, cols, eps, relu, p, seed, offset, dev_offset, tee, si));
    return check_launch("egk_rowln_fwd");
}

int egk_rowln_bwd_ws_rows(int32_t rows) { return row_grid(rows); }

static int fill_ln_groups(const char* what, RowLNGroups& G, const float* const* w, const float* const* b, const int32_t* row_ptr,
                          int32_t n_groups, int& max_rows) {
    EGK_REQUIRE(w && b && row_ptr && n_groups >= 1 && n_groups <= LN_MAX_GROUPS, "%s: 1 .. %d row ranges", what, LN_MAX_GROUPS);
    max_rows = 0;
    for (int g = 0; g < LN_MAX_GROUPS; ++g) {
        const int k = g < n_groups ? g : n_groups - 1;
        EGK_REQUIRE(w[k] && b[k], "%s: null parameter pointer", what);
        G.w[g] = w[k]; G.b[g] = b[k];
    }
    for (int g = 0; g <= LN_MAX_GROUPS; ++g) G.row0[g] = row_ptr[g <= n_groups ? g : n_groups];
    for (int g = 0; g < n_groups; ++g) {
        EGK_REQUIRE(row_ptr[g + 1] >= row_ptr[g], "%s: row_ptr must be non-decreasing", what);
        max_rows = row_ptr[g + 1] - row_ptr[g] > max_rows ? row_ptr[g + 1] - row_ptr[g] : max_rows;
    }
    return 0;
}

int egk_rowln_group_fwd(egk_stream_t stream, const void* x, const float* const* w, const float* const* b, const int32_t* row_ptr,
                        int32_t n_groups, void* y, float* mean, float* rstd, int32_t cols, float eps, int32_t relu,
                        int32_t dtype) {
    EGK_REQUIRE(x && y && mean && rstd, "egk_rowln_group_fwd: null pointer");
    RowLNGroups G;
    int max_rows;
    const int rc = fill_ln_groups("egk_rowln_group_fwd", G, w, b, row_ptr, n_groups, max_rows);
    if (rc) return rc;
    if (max_rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    ProfScope prof(KID_ROWLN_FWD, s, 0, 2 * eb * (row_ptr[n_groups] - row_ptr[0]) * cols);
    const int per = cdiv(row_grid_wide(row_ptr[n_groups] - row_ptr[0]), n_groups);
    const SplitTee tee = take_split_tee();  // (the halves are indexed by the row of the shared [rows, cols] buffer, as y is)
    EGK_REQUIRE(!tee.lo || dtype == EGK_F32, "egk_rowln_group_fwd: a split tee needs an f32 result");
    DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((rowln_fwd_group_kernel<NV, T, FULL>), dim3(per < 1 ? 1 : per, n_groups), dim3(256), 0,
                                                 s, (const T*)x, G, (T*)y, mean, rstd, cols, eps, relu, tee));
    return check_launch("egk_rowln_group_fwd");
}

/* partial rows of range g: ws[(g * egk_rowln_bwd_ws_rows(max range rows) + block)][2][cols]; reduce each range with
 * egk_ln_bwd_reduce(ws + g * blocks * 2 * cols, dw_g, db_g, max range rows, cols, 0) */
int egk_rowln_group_bwd(egk_stream_t stream, const void* dy, const void* x, const float* const* w, const float* const* b,
                        const int32_t* row_ptr, int32_t n_groups, const float* mean, const float* rstd, void* dx, float* ws,
                        int32_t cols, int32_t relu, int32_t dtype) {
    EGK_REQUIRE(dy && x && mean && rstd && dx && ws, "egk_rowln_group_bwd: null pointer");
    RowLNGroups G;
    int max_rows;
    const int rc = fill_ln_groups("egk_rowln_group_bwd", G, w, b, row_ptr, n_groups, max_rows);
    if (rc) return rc;
    if (max_rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(max_rows);
    ProfScope prof(KID_ROWLN_BWD, s, 0, (dtype == EGK_BF16 ? 6.0 : 12.0) * (row_ptr[n_groups] - row_ptr[0]) * cols);
    DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((rowln_bwd_group_kernel<NV, T, FULL>), dim3(grid, n_groups), dim3(256),
                                                 WPB * 2 * NV * 256 * sizeof(float), s, (const T*)dy, (const T*)x, G, mean, rstd,
                                                 (T*)dx, ws, cols, relu));
    return check_launch("egk_rowln_group_bwd");
}

int egk_rowln_bwd(egk_stream_t stream, const void* dy, const void* x, const float* w, const float* b, const float* mean,
                  const float* rstd, const uint8_t* mask, void* dx, float* dw, float* db, float* ws, int32_t rows,
                  int32_t cols, int32_t relu, float p, int32_t dtype) {
    EGK_REQUIRE(dy && x && w && b && mean && rstd && dx && ws, "egk_rowln_bwd: null pointer");  // (dw = db = NULL: see egk_ln_bwd_reduce)
    EGK_REQUIRE(p == 0.f || mask, "egk_rowln_bwd: dropout needs the mask");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows);
    {
        ProfScope prof(KID_ROWLN_BWD, s, 0, (dtype == EGK_BF16 ? 6.0 : 12.0) * rows * cols);
        if (wide_rows_ok(cols, dy, dx, w, b, mask, dtype) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
            EGK_DISPATCH_T(dtype, DISPATCH_NVW_(cols, hipLaunchKernelGGL((rowln_bwd_wide_kernel<NVW, T>), dim3(grid), dim3(256), 0, s,
                                                                        (const T*)dy, (const T*)x, w, b, mean, rstd, mask, (T*)dx, ws,
                                                                        rows, relu, p)));
        } else {
            DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((rowln_bwd_kernel<NV, T, FULL>), dim3(grid), dim3(256),
                                                         WPB * 2 * NV * 256 * sizeof(float), s, (const T*)dy, (const T*)x, w, b, mean,
                                                         rstd, mask, (T*)dx, ws, rows, cols, relu, p));
        }
    }
    if (dw || db) {
        ProfScope prof(KID_ROWLN_BWD_REDUCE, s, 0, 8.0 * grid * cols);
        hipLaunchKernelGGL(partial_reduce2_kernel, dim3(cdiv(cols, RED_COLS)), dim3(1024), 0, s, ws, dw, db, grid, cols);
    }
    return check_launch("egk_rowln_bwd");
}

// The parameter-gradient half of the two backward entry points above, for callers that passed dw = db = NULL there and
// want the reduction of the per-workgroup partial rows on another stream (it feeds nothing but the optimizer).
int egk_ln_bwd_reduce(egk_stream_t stream, const void* ws, float* dw, float* db, int32_t rows, int32_t cols, int32_t n_seg) {
    EGK_REQUIRE(ws && dw && db, "egk_ln_bwd_reduce: null pointer");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows);
    // n_seg = 0: egk_rowln_bwd's workspace (partial rows first); n_seg >= 1: egk_graphln_bwd's (segment sums first)
    const float* ws_col = n_seg > 0 ? (const float*)((const char*)ws + (int64_t)grid * n_seg * 2 * 8) : (const float*)ws;
    ProfScope prof(n_seg > 0 ? KID_GRAPHLN_BWD_REDUCE : KID_ROWLN_BWD_REDUCE, s, 0, 8.0 * grid * cols);
    hipLaunchKernelGGL(partial_reduce2_kernel, dim3(cdiv(cols, RED_COLS)), dim3(1024), 0, s, ws_col, dw, db, grid, cols);
    return check_launch("egk_ln_bwd_reduce");
}

/* ``count`` (<= 8) reductions egk_ln_bwd_reduce(ws[i], dw[i], db[i], rows[i], cols[i], n_seg[i]) in one launch. */
int egk_ln_bwd_reduce_multi(egk_stream_t stream, const void* const* ws, float* const* dw, float* const* db, const int32_t* rows,
                            const int32_t* cols, const int32_t* n_seg, int32_t count) {
    EGK_REQUIRE(ws && dw && db && rows && cols && n_seg && count >= 1 && count <= REDUCE_MAX, "egk_ln_bwd_reduce_multi: 1 .. %d reductions",
                REDUCE_MAX);
    ReduceBatch B;
    int max_cols = 0;
    double bytes = 0;
    for (int i = 0; i < REDUCE_MAX; ++i) {
        const int k = i < count ? i : count - 1;
        EGK_REQUIRE(ws[k] && dw[k] && db[k], "egk_ln_bwd_reduce_multi: null pointer");
        const int grid = row_grid(rows[k]);
        B.ws[i] = n_seg[k] > 0 ? (const float*)((const char*)ws[k] + (int64_t)grid * n_seg[k] * 2 * 8) : (const float*)ws[k];
        B.oa[i] = dw[k]; B.ob[i] = db[k]; B.nblk[i] = rows[k] > 0 ? grid : 0; B.cols[i] = cols[k];
        if (i < count) {
            max_cols = cols[k] > max_cols ? cols[k] : max_cols;
            bytes += 8.0 * grid * cols[k];
        }
    }
    if (max_cols == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_ROWLN_BWD_REDUCE, s, 0, bytes);
    hipLaunchKernelGGL(partial_reduce2_multi_kernel, dim3(cdiv(max_cols, RED_COLS), count), dim3(1024), 0, s, B);
    return check_launch("egk_ln_bwd_reduce_multi");
}

int64_t egk_graphln_ws_bytes(int32_t rows, int32_t cols, int32_t n_seg) {
    const int64_t g = row_grid(rows);
    return g * n_seg * 2 * 8 + g * 2 * (int64_t)cols * 4;
}

int egk_graphln_fwd(egk_stream_t stream, const void* x, const float* w, const float* b, void* y, float* stats,
                    const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols, float eps, float slope, void* ws,
                    int32_t dtype) {
    EGK_REQUIRE(x && w && b && y && stats && seg_ptr && ws, "egk_graphln_fwd: null pointer");
    EGK_REQUIRE(n_seg >= 1 && n_seg <= MAXSEG, "egk_graphln_fwd: n_seg must be in [1,%d]", MAXSEG);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows);
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    // (a slab input, egk_slab_input_next: the statistics launch reads the two slabs and leaves the reduced matrix in x_out, which the
    //  normalising launch then reads)
    const SlabInput si = take_slab_input();
    EGK_REQUIRE(!si.x2 || (dtype == EGK_F32 && cols % 4 == 0 && cols <= 1024), "egk_graphln_fwd: a slab input needs f32 rows of <= 1024 columns");
    {
        ProfScope prof(KID_GRAPHLN_STATS, s, 0, eb * rows * cols);
        DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((graphln_stats_kernel<NV, T, FULL>), dim3(grid), dim3(256), 0, s, (const T*)x, seg_ptr,
                                                     n_seg, rows, cols, (double*)ws, si));
    }
    const void* xn = si.x2 ? (const void*)si.x_out : x;
    {
        ProfScope prof(KID_GRAPHLN_FWD, s, 0, 2 * eb * rows * cols);
        const SplitTee tee = take_split_tee();
        EGK_REQUIRE(!tee.lo || dtype == EGK_F32, "egk_graphln_fwd: a split tee needs an f32 result");
        DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((graphln_fwd_kernel<NV, T, FULL>), dim3(row_grid_wide(rows)), dim3(256), 0, s, (const T*)xn, w, b,
                                                     (T*)y, stats, seg_ptr, n_seg, rows, cols, eps, slope, (const double*)ws, grid, tee));
    }
    return check_launch("egk_graphln_fwd");
}

/* The normalising pass alone, from per-block segment sums computed elsewhere: ``partials`` = double [n_partials][n_seg][2]
 * (sum, sum of squares of x per segment), e.g. written by the epilogue of the contraction that produced x
 * (egk_gemm_desc.st_mode 1).  Same arithmetic as egk_graphln_fwd's second launch. */
int egk_graphln_fwd_apply(egk_stream_t stream, const void* x, const float* w, const float* b, void* y, float* stats,
                          const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols, float eps, float slope,
                          const void* partials, int32_t n_partials, int32_t dtype) {
    EGK_REQUIRE(x && w && b && y && stats && seg_ptr && partials, "egk_graphln_fwd_apply: null pointer");
    EGK_REQUIRE(n_seg >= 1 && n_seg <= MAXSEG && n_partials >= 1, "egk_graphln_fwd_apply: bad segment / partial count");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    ProfScope prof(KID_GRAPHLN_FWD, s, 0, 2 * eb * rows * cols);
    const SplitTee tee = take_split_tee();
    EGK_REQUIRE(!tee.lo || dtype == EGK_F32, "egk_graphln_fwd_apply: a split tee needs an f32 result");
    DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((graphln_fwd_kernel<NV, T, FULL>), dim3(row_grid_wide(rows)), dim3(256), 0, s, (const T*)x, w, b,
                                                 (T*)y, stats, seg_ptr, n_seg, rows, cols, eps, slope, (const double*)partials, n_partials, tee));
    return check_launch("egk_graphln_fwd_apply");
}

/* Backward from per-block segment sums computed elsewhere: ``partials`` = double [n_partials][n_seg][2] holding
 * (sum dxhat, sum dxhat * xhat) per segment (egk_gemm_desc.st_mode 2: the epilogue of the contraction that produced dy).
 * Writes dx and the per-workgroup partial rows of dw / db to ws_col (f32 [egk_rowln_bwd_ws_rows(rows)][2][cols]; reduce
 * with egk_ln_bwd_reduce(ws_col, dw, db, rows, cols, 0)). */
int egk_graphln_bwd_apply(egk_stream_t stream, const void* dy, const void* x, const float* w, const float* b, const float* stats,
                          void* dx, const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols, float eps, float slope,
                          const void* partials, int32_t n_partials, float* ws_col, int32_t dtype) {
    EGK_REQUIRE(dy && x && w && b && stats && dx && seg_ptr && partials && ws_col, "egk_graphln_bwd_apply: null pointer");
    EGK_REQUIRE(n_seg >= 1 && n_seg <= MAXSEG && n_partials >= 1, "egk_graphln_bwd_apply: bad segment / partial count");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    ProfScope prof(KID_GRAPHLN_BWD, s, 0, 3 * eb * rows * cols);
    DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((graphln_bwd_kernel<NV, T, FULL, true>), dim3(row_grid(rows)), dim3(256),
                                                 WPB * 2 * NV * 256 * sizeof(float), s, (const T*)dy, (const T*)x, w, b, stats, (T*)dx,
                                                 seg_ptr, n_seg, rows, cols, eps, slope, (const double*)partials, n_partials, ws_col));
    return check_launch("egk_graphln_bwd_apply");
}

/* The three launches of egk_graphln_bwd as two calls, so that the segment sums can be combined with those of other
 * ranks in between (exact cross-rank statistics, egopack_amd/ops.py graph_ln_exchange): egk_graphln_bwd_stats writes the
 * per-block (sum dxhat, sum dxhat * xhat) per segment to the head of ws (double [egk_graphln_stats_blocks(rows)][n_seg][2])
 * and the dw / db partial rows behind them; egk_graphln_bwd_finish normalises with the sums in ``partials`` (the head of
 * ws itself, or any double [n_partials][n_seg][2]) and reduces the dw / db rows of ws. */
int32_t egk_graphln_stats_blocks(int32_t rows) { return row_grid(rows); }

int egk_graphln_stats(egk_stream_t stream, const void* x, const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols,
                      void* partials, int32_t dtype) {
    EGK_REQUIRE(x && seg_ptr && partials, "egk_graphln_stats: null pointer");
    EGK_REQUIRE(n_seg >= 1 && n_seg <= MAXSEG, "egk_graphln_stats: n_seg must be in [1,%d]", MAXSEG);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    ProfScope prof(KID_GRAPHLN_STATS, s, 0, eb * rows * cols);
    DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((graphln_stats_kernel<NV, T, FULL>), dim3(row_grid(rows)), dim3(256), 0, s, (const T*)x,
                                                 seg_ptr, n_seg, rows, cols, (double*)partials));
    return check_launch("egk_graphln_stats");
}

int egk_graphln_bwd_stats(egk_stream_t stream, const void* dy, const void* x, const float* w, const float* b, const float* stats,
                          const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols, float slope, void* ws, int32_t dtype) {
    EGK_REQUIRE(dy && x && w && b && stats && seg_ptr && ws, "egk_graphln_bwd_stats: null pointer");
    EGK_REQUIRE(n_seg >= 1 && n_seg <= MAXSEG, "egk_graphln_bwd_stats: n_seg must be in [1,%d]", MAXSEG);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows);
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    double* ws_seg = (double*)ws;
    float* ws_col = (float*)((char*)ws + (int64_t)grid * n_seg * 2 * 8);
    ProfScope prof(KID_GRAPHLN_BWD_STATS, s, 0, 2 * eb * rows * cols);
    DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((graphln_bwd_stats_kernel<NV, T, FULL>), dim3(grid), dim3(256),
                                                 WPB * 2 * NV * 256 * sizeof(float), s, (const T*)dy, (const T*)x, w, b, stats,
                                                 seg_ptr, n_seg, rows, cols, slope, ws_seg, ws_col));
    return check_launch("egk_graphln_bwd_stats");
}

int egk_graphln_bwd_finish(egk_stream_t stream, const void* dy, const void* x, const float* w, const float* b, const float* stats,
                           void* dx, float* dw, float* db, const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols,
                           float eps, float slope, const void* partials, int32_t n_partials, const void* ws, int32_t dtype) {
    EGK_REQUIRE(dy && x && w && b && stats && dx && seg_ptr && partials && ws, "egk_graphln_bwd_finish: null pointer");
    EGK_REQUIRE(n_seg >= 1 && n_seg <= MAXSEG && n_partials >= 1, "egk_graphln_bwd_finish: bad segment / partial count");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows);
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    const float* ws_col = (const float*)((const char*)ws + (int64_t)grid * n_seg * 2 * 8);
    {
        ProfScope prof(KID_GRAPHLN_BWD, s, 0, 3 * eb * rows * cols);
        DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((graphln_bwd_kernel<NV, T, FULL>), dim3(row_grid_wide(rows)), dim3(256), 0, s, (const T*)dy,
                                                     (const T*)x, w, b, stats, (T*)dx, seg_ptr, n_seg, rows, cols, eps, slope,
                                                     (const double*)partials, n_partials));
    }
    if (dw || db) {
        ProfScope prof(KID_GRAPHLN_BWD_REDUCE, s, 0, 8.0 * grid * cols);
        hipLaunchKernelGGL(partial_reduce2_kernel, dim3(cdiv(cols, RED_COLS)), dim3(1024), 0, s, ws_col, dw, db, grid, cols);
    }
    return check_launch("egk_graphln_bwd_finish");
}

int egk_graphln_bwd(egk_stream_t stream, const void* dy, const void* x, const float* w, const float* b,
                    const float* stats, void* dx, float* dw, float* db, const int32_t* seg_ptr, int32_t n_seg,
                    int32_t rows, int32_t cols, float eps, float slope, void* ws, int32_t dtype) {
    EGK_REQUIRE(dy && x && w && b && stats && dx && seg_ptr && ws, "egk_graphln_bwd: null pointer");
    if (rows == 0) return 0;
    const int rc = egk_graphln_bwd_stats(stream, dy, x, w, b, stats, seg_ptr, n_seg, rows, cols, slope, ws, dtype);
    if (rc) return rc;
    return egk_graphln_bwd_finish(stream, dy, x, w, b, stats, dx, dw, db, seg_ptr, n_seg, rows, cols, eps, slope, ws, row_grid(rows),
                                  ws, dtype);
}
/* One-logit classifier + BCE-with-logits over the rows of f, loss AND gradients in one pass (see rowdot_bce_kernel).
 * w: the classifier's weight row in the element type of f (the operand copy the contraction path would read).
 * df == NULL: forward only (logits, loss).  Otherwise df = g w per row and ws = float [egk_rowdot_ws_rows(rows)][cols + 4]
 * receives the partial rows of dw / db; egk_rowdot_reduce ACCUMULATES them into dw [cols] and db [1]. */
int32_t egk_rowdot_ws_rows(int32_t rows) { return row_grid(rows); }

int egk_rowdot_bce(egk_stream_t stream, const void* f, const void* w, const float* bias, const int64_t* y, float* logits,
                   float* loss, void* df, float* ws, int32_t rows, int32_t cols, float seed, int32_t dtype) {
    EGK_REQUIRE(f && w && y && logits && loss, "egk_rowdot_bce: null pointer");
    EGK_REQUIRE(!df || ws, "egk_rowdot_bce: gradients need the partial-row workspace");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    ProfScope prof(KID_BCE_FWD, s, 2.0 * rows * cols * (df ? 3 : 1), eb * rows * cols * (df ? 2 : 1) + 16.0 * rows);
    DISPATCH_NV(cols, dtype, hipLaunchKernelGGL((rowdot_bce_kernel<NV, T, FULL>), dim3(row_grid(rows)), dim3(256),
                                                 WPB * (NV * 256 + 4) * sizeof(float), s, (const T*)f,
                                                 (const T*)w, bias, (const long long*)y, logits, loss, (T*)df, ws, rows, cols, seed));
    return check_launch("egk_rowdot_bce");
}

int32_t egk_rowdot_ce2_max_rows(void) { return CE2_MAX_ROWS; }

int egk_rowdot_ce2_multi(egk_stream_t stream, int32_t n_src, const void* const* f, const void* const* w, const float* const* bias,
                         const int64_t* y, float* logits, float* loss, void* const* df, float* const* dw, float* const* db,
                         float* gws, int32_t rows, int32_t cols, int32_t average, float smoothing, float seed, int32_t dtype) {
    // seed < 0 is not a gradient scale: the sign bit of ``average`` is not available either -- the phase rides in bits 1-2 of ``average``
    const int phase = (average >> 1) & 3;  // 0: both launches; 1: the row launch only; 2: the column launch only (gws from a phase-1 call)
    average &= 1;
    EGK_REQUIRE(f && w && y && logits && loss && n_src >= 1 && n_src <= CE2_MAX_SRC, "egk_rowdot_ce2_multi: 1 .. %d sources", CE2_MAX_SRC);
    EGK_REQUIRE(rows >= 0 && rows <= CE2_MAX_ROWS && cols >= 1, "egk_rowdot_ce2_multi: at most %d rows", CE2_MAX_ROWS);
    if (rows == 0) return 0;
    CE2Sources S;
    bool want = false, want_w = false;
    for (int k = 0; k < CE2_MAX_SRC; ++k) {
        const bool in = k < n_src;
        S.f[k] = in ? f[k] : nullptr;
        S.w[k] = in ? w[k] : nullptr;
        S.bias[k] = (in && bias) ? bias[k] : nullptr;
        S.df[k] = (in && df) ? df[k] : nullptr;
        S.dw[k] = (in && dw) ? dw[k] : nullptr;
        S.db[k] = (in && db) ? db[k] : nullptr;
        EGK_REQUIRE(!in || (S.f[k] && S.w[k]), "egk_rowdot_ce2_multi: null source");
        want = want || S.df[k] || S.dw[k] || S.db[k];
        want_w = want_w || S.dw[k] || S.db[k];
    }
    S.n = n_src;
    EGK_REQUIRE(!want || gws, "egk_rowdot_ce2_multi: gradients need the [rows][2] workspace");
    hipStream_t s = (hipStream_t)stream;
    const double eb = dtype == EGK_BF16 ? 2.0 : 4.0;
    ProfScope prof(KID_CE_FWD, s, 4.0 * rows * cols * n_src * (want ? 3 : 1), eb * rows * cols * n_src * (want ? 3 : 1));
    EGK_DISPATCH_T(dtype, {
        if (phase != 2)
            hipLaunchKernelGGL(rowdot_ce2_rows_kernel<T>, dim3(rows), dim3(256), 0, s, S, (const long long*)y, logits, loss,
                               want ? gws : nullptr, cols, average ? 1.f / n_src : 1.f, smoothing, seed);
        if (want_w && phase != 1) hipLaunchKernelGGL(rowdot_ce2_cols_kernel<T>, dim3(cdiv(cols, 256), n_src), dim3(64), 0, s, S, gws, rows, cols);
    });
    return check_launch("egk_rowdot_ce2_multi");
}

int egk_rowdot_ce2(egk_stream_t stream, const void* f, const void* w, const float* bias, const int64_t* y, float* logits, float* loss,
                   void* df, float* dw, float* db, float* gws, int32_t rows, int32_t cols, float smoothing, float seed, int32_t dtype) {
    EGK_REQUIRE(f && w && y && logits && loss, "egk_rowdot_ce2: null pointer");
    EGK_REQUIRE(!df || dw, "egk_rowdot_ce2: gradients need dw");
    return egk_rowdot_ce2_multi(stream, 1, &f, &w, &bias, y, logits, loss, &df, &dw, &db, gws, rows, cols, 0, smoothing, seed, dtype);
}

int egk_rowdot_reduce(egk_stream_t stream, const float* ws, float* dw, float* db, int32_t rows, int32_t cols) {
    EGK_REQUIRE(ws && dw, "egk_rowdot_reduce: null pointer");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows);
    ProfScope prof(KID_ROWLN_BWD_REDUCE, s, 0, 4.0 * grid * (cols + 4));
    hipLaunchKernelGGL(rowdot_reduce_kernel, dim3(cdiv(cols + 1, 16)), dim3(256), 0, s, ws, dw, db, grid, cols);
    return check_launch("egk_rowdot_reduce");
}
}
