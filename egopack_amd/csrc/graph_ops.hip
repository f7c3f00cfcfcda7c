// Message-passing and prototype kernels: positional encoding add, CSR row gather (SAGE mean
// forward/backward), GraphONE gather-max, per-sequence max pool, cosine / L2 k-NN selection and the
// float64 prototype-bank accumulation (label-grouped, no atomics).
//
// All row kernels use the one-wave-per-row layout (16 B per lane, 1 KiB per wave-instruction):
// a 1024-wide fp32 feature row is 4 wave-instructions.  Neighbour rows of the banded temporal
// graph and the prototype bank (K*H*4 = 16.8 MB at K = 4096) are re-read from L2 / Infinity
// Cache, so the algorithmic HBM bytes are one read + one write of the [N, H] tensor.
#include <math.h>

#include "common.h"
#include "rows1024.h"

namespace egk {

constexpr int WPB = 4;

#define ld4 ld4t
#define st4 st4t

// ---- positional encoding ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pe_add_kernel(const T* __restrict__ x, const long long* __restrict__ pos,
                                                     const float* __restrict__ freq, T* __restrict__ y, int rows,
                                                     int cols, const SplitTee tee) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (cols & 3) == 0;
    const int half = cols >> 1;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const float p = (float)pos[row];
        const T* xr = x + (long long)row * cols;
        T* yr = y + (long long)row * cols;
        for (int c = lane * 4; c < cols; c += 256) {
            float4 v = ld4(xr, c, cols, vec);
            float e[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cc = c + t;
                if (cc < cols) {
                    const bool is_sin = cc < half;
                    const float a = p * freq[is_sin ? cc : cc - half];
                    e[t] = is_sin ? sinf(a) : cosf(a);
                } else
                    e[t] = 0.f;
            }
            v.x += e[0]; v.y += e[1]; v.z += e[2]; v.w += e[3];
            st4(yr, c, cols, vec, v);
            tee4(tee, row, c, cols, vec, v);
        }
    }
}

// PE(p)[c] for every integer position p in [pos_min, pos_min + n_pos): the same sinf / cosf evaluations as pe_add_kernel, once
// per (position, column) instead of once per (node, column) -- T = 32 positions against 6144 nodes in the headline step.
__global__ __launch_bounds__(256) void pe_table_kernel(const float* __restrict__ freq, long long pos_min, int n_pos, int cols,
                                                       float* __restrict__ table) {
    const int half = cols >> 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)n_pos * cols; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / cols), cc = (int)(i - (long long)r * cols);
        const float p = (float)(pos_min + r);
        const bool is_sin = cc < half;
        const float a = p * freq[is_sin ? cc : cc - half];
        table[i] = is_sin ? sinf(a) : cosf(a);
    }
}

// y = x + table[pos - pos_min] (rows of the table come from L2); positions outside the table are evaluated directly
template <typename T>
__global__ __launch_bounds__(256) void pe_add_table_kernel(const T* __restrict__ x, const long long* __restrict__ pos,
                                                           const float* __restrict__ freq, const float* __restrict__ table,
                                                           long long pos_min, int n_pos, T* __restrict__ y, int rows, int cols,
                                                           const SplitTee tee, const SlabInput si = SlabInput{nullptr, nullptr, nullptr}) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (cols & 3) == 0;
    const int half = cols >> 1;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const long long pr = pos[row];
        const long long ti = pr - pos_min;
        const bool hit = ti >= 0 && ti < n_pos;  // (wave-uniform)
        const float* tr = table + (hit ? ti : 0) * cols;
        const T* xr = x + (long long)row * cols;
        T* yr = y + (long long)row * cols;
        for (int c = lane * 4; c < cols; c += 256) {
            float4 v = ld4(xr, c, cols, vec);
            if constexpr (sizeof(T) == 4) {
                if (si.x2) {  // the input as two K slabs + bias (egk_slab_input_next): gemm_splitk_reduce's arithmetic, stored for later readers
                    const float4 v2 = ld4t(si.x2 + (long long)row * cols, c, cols, vec);
                    v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
                    if (si.bias) {
                        const float4 bb = ld4t(si.bias, c, cols, vec);
                        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                    }
                    st4t(si.x_out + (long long)row * cols, c, cols, vec, v);
                }
            }
            float4 e;
            if (hit) {
                e = ld4t(tr, c, cols, vec);
            } else {
                float q[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int cc = c + t;
                    q[t] = 0.f;
                    if (cc < cols) {
                        const bool is_sin = cc < half;
                        const float a = (float)pr * freq[is_sin ? cc : cc - half];
                        q[t] = is_sin ? sinf(a) : cosf(a);
                    }
                }
                e = make_float4(q[0], q[1], q[2], q[3]);
            }
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
            st4(yr, c, cols, vec, v);
            tee4(tee, row, c, cols, vec, v);
        }
    }
}

// ---- CSR gather ----------------------------------------------------------------------------------
// One wave per output row, the row in NV float4 registers per lane.  Neighbour indices / weights are fetched by
// the lanes in one load per 64 edges and broadcast with shuffles; 4 neighbour rows are in flight per step.
// HEAVY rows (more than HEAVY edges: the LTA fan-out node has out-degree 31 in the backward orientation) are not
// walked by their wave alone: after the light rows of a workgroup are done, its 4 waves share each heavy row's
// edge list (edge e -> wave e % 4), combine their partial rows through LDS in wave order and wave 0 stores.
constexpr int HEAVY = 12;

template <int NV, typename T>
__device__ __forceinline__ void csr_accumulate(const T* __restrict__ x, const int* __restrict__ col, const float* __restrict__ wgt,
                                               int e0, int e1, int e_first, int e_stride, int cols, bool vec, int lane,
                                               float4 (&acc)[NV], long long stride = -1) {
    if (stride < 0) stride = cols;  // (a column slice of wider rows passes the row stride and its own width as cols)
    // edges e0 + e_first, e0 + e_first + e_stride, ... < e1
    const int n_mine = (e1 - e0 - e_first + e_stride - 1) / e_stride;
    for (int base = 0; base < n_mine; base += 64) {
        const int cnt = min(64, n_mine - base);
        const int my_e = e0 + e_first + (base + lane) * e_stride;
        const int my_c = lane < cnt ? col[my_e] : 0;
        const float my_w = (wgt && lane < cnt) ? wgt[my_e] : 1.f;
        int e = 0;
        if constexpr (NV == 1) {
            // a column slice of a listed heavy row (one float4 per lane per neighbour row): 16 neighbour rows in flight --
            // with 4 the walk over the fan-out node's 31 edges is 8 dependent round trips, with 16 it is 2.  The rows are
            // still ADDED in edge order.
            for (; e + 16 <= cnt; e += 16) {
                float4 v[16];
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const T* su = x + (long long)__shfl(my_c, e + u, 64) * stride;
                    w[u] = __shfl(my_w, e + u, 64);
                    v[u] = ld4(su, lane * 4, cols, vec);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    acc[0].x += w[u] * v[u].x; acc[0].y += w[u] * v[u].y; acc[0].z += w[u] * v[u].z; acc[0].w += w[u] * v[u].w;
                }
            }
        }
        for (; e + 4 <= cnt; e += 4) {
            const T* s0 = x + (long long)__shfl(my_c, e + 0, 64) * stride;
            const T* s1 = x + (long long)__shfl(my_c, e + 1, 64) * stride;
            const T* s2 = x + (long long)__shfl(my_c, e + 2, 64) * stride;
            const T* s3 = x + (long long)__shfl(my_c, e + 3, 64) * stride;
            const float w0 = __shfl(my_w, e + 0, 64), w1 = __shfl(my_w, e + 1, 64);
            const float w2 = __shfl(my_w, e + 2, 64), w3 = __shfl(my_w, e + 3, 64);
            float4 v0[NV], v1[NV], v2[NV], v3[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                v0[i] = ld4(s0, c, cols, vec); v1[i] = ld4(s1, c, cols, vec);
                v2[i] = ld4(s2, c, cols, vec); v3[i] = ld4(s3, c, cols, vec);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                acc[i].x += w0 * v0[i].x; acc[i].y += w0 * v0[i].y; acc[i].z += w0 * v0[i].z; acc[i].w += w0 * v0[i].w;
                acc[i].x += w1 * v1[i].x; acc[i].y += w1 * v1[i].y; acc[i].z += w1 * v1[i].z; acc[i].w += w1 * v1[i].w;
                acc[i].x += w2 * v2[i].x; acc[i].y += w2 * v2[i].y; acc[i].z += w2 * v2[i].z; acc[i].w += w2 * v2[i].w;
                acc[i].x += w3 * v3[i].x; acc[i].y += w3 * v3[i].y; acc[i].z += w3 * v3[i].z; acc[i].w += w3 * v3[i].w;
            }
        }
        for (; e < cnt; ++e) {
            const T* src = x + (long long)__shfl(my_c, e, 64) * stride;
            const float we = __shfl(my_w, e, 64);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 v = ld4(src, (i * 64 + lane) * 4, cols, vec);
                acc[i].x += we * v.x; acc[i].y += we * v.y; acc[i].z += we * v.z; acc[i].w += we * v.w;
            }
        }
    }
}

template <int NV, typename T>
__device__ __forceinline__ void csr_finish(float4 (&acc)[NV], const float* __restrict__ wgt, float mean_w, const T* __restrict__ gate,
                                           T* __restrict__ out, int row, int cols, bool vec, int lane, long long stride = -1,
                                           const SplitTee tee = SplitTee{nullptr, nullptr, 0}) {
    if (stride < 0) stride = cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (!wgt) {  // mean = sum / count, as scatter_add / clamp(count, 1)
            acc[i].x *= mean_w; acc[i].y *= mean_w; acc[i].z *= mean_w; acc[i].w *= mean_w;
        }
        if (gate) {
            const float4 g = ld4(gate + (long long)row * stride, c, cols, vec);
            acc[i].x = g.x > 0.f ? acc[i].x : 0.f; acc[i].y = g.y > 0.f ? acc[i].y : 0.f;
            acc[i].z = g.z > 0.f ? acc[i].z : 0.f; acc[i].w = g.w > 0.f ? acc[i].w : 0.f;
        }
        st4(out + (long long)row * stride, c, cols, vec, acc[i]);
        tee4(tee, row, c, cols, vec, acc[i]);  // (a column slice hands in halves advanced by its first column)
    }
}

template <int NV, typename T>
__global__ __launch_bounds__(256) void csr_gather_kernel(const T* __restrict__ x, const int* __restrict__ rowptr,
                                                         const int* __restrict__ col, const float* __restrict__ wgt,
                                                         const T* __restrict__ gate, T* __restrict__ out, int rows,
                                                         int cols, int skip_above, const int* __restrict__ block_rows,
                                                         int n_block_rows, const unsigned char* __restrict__ band = nullptr,
                                                         const SplitTee tee = SplitTee{nullptr, nullptr, 0}) {
    extern __shared__ __attribute__((aligned(16))) float part[];  // [3][NV*256] partial rows of waves 1..3
    __shared__ int heavy[64];
    __shared__ int n_heavy;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (cols & 3) == 0;
    if ((int)blockIdx.x < n_block_rows) {
        // One of the listed rows (more than skip_above edges, at most a few dozen: the LTA fan-out node at T = 32), summed
        // by THIS workgroup alone while the others do the light rows: every wave walks all the edges, in order, for its
        // quarter of the columns -- the summation order of a light row, no partial rows, no second launch.
        constexpr int NVW = NV >= 4 ? NV / 4 : 1;
        const int row = block_rows[blockIdx.x], c0 = wave * NVW * 256;
        if (c0 >= cols) return;
        const int e0 = rowptr[row], e1 = rowptr[row + 1];
        float4 acc[NVW];
#pragma unroll
        for (int i = 0; i < NVW; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        csr_accumulate<NVW, T>(x + c0, col, wgt, e0, e1, 0, 1, cols - c0, vec, lane, acc, cols);
        csr_finish<NVW, T>(acc, wgt, e1 > e0 ? 1.f / (float)(e1 - e0) : 0.f, gate ? gate + c0 : gate, out + c0, row, cols - c0, vec,
                           lane, cols, SplitTee{tee.lo ? tee.hi + c0 : nullptr, tee.lo ? tee.lo + c0 : nullptr, tee.ld});
        return;
    }
    const int bid = blockIdx.x - n_block_rows, nblk = gridDim.x - n_block_rows;
    if (threadIdx.x == 0) n_heavy = 0;
    __syncthreads();
    const RowWalk rw = (n_block_rows & 7) == 0 ? row_walk(bid, nblk, 0, rows, wave, WPB) : RowWalk{bid * WPB + wave, rows, nblk * WPB};
    for (int row = rw.first; row < rw.end; row += rw.step) {
        if (band) {
            // BANDED row (data.build_csr): its neighbours are a subset of {row - 1, row, row + 1}, named by three bits -- no
            // rowptr / col fetches, the neighbour rows are requested at once.  Added in ascending order, which is the order
            // of such a row's CSR entries (the builder checks): the same sums, bit for bit.  0xFF: a general row, below.
            const unsigned code = band[row];
            if (code != 0xFFu) {
                float4 acc[NV], va[NV], vb[NV], vc[NV];
                const long long r0 = (long long)row * cols;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = (i * 64 + lane) * 4;
                    acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    va[i] = (code & 1u) ? ld4(x + r0 - cols, c, cols, vec) : acc[i];
                    vb[i] = (code & 2u) ? ld4(x + r0, c, cols, vec) : acc[i];
                    vc[i] = (code & 4u) ? ld4(x + r0 + cols, c, cols, vec) : acc[i];
                }
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (code & 1u) { acc[i].x += 1.f * va[i].x; acc[i].y += 1.f * va[i].y; acc[i].z += 1.f * va[i].z; acc[i].w += 1.f * va[i].w; }
                    if (code & 2u) { acc[i].x += 1.f * vb[i].x; acc[i].y += 1.f * vb[i].y; acc[i].z += 1.f * vb[i].z; acc[i].w += 1.f * vb[i].w; }
                    if (code & 4u) { acc[i].x += 1.f * vc[i].x; acc[i].y += 1.f * vc[i].y; acc[i].z += 1.f * vc[i].z; acc[i].w += 1.f * vc[i].w; }
                }
                const int cnt = __popc(code & 7u);
                csr_finish<NV, T>(acc, nullptr, cnt ? 1.f / (float)cnt : 0.f, gate, out, row, cols, vec, lane, -1, tee);
                continue;
            }
        }
        const int e0 = rowptr[row], e1 = rowptr[row + 1];
        if (e1 - e0 > skip_above) continue;  // listed by the host: the split launches below produce this row
        if (e1 - e0 > HEAVY) {  // deferred to the cooperative phase (wave-uniform branch) while the list has room
            int slot = lane == 0 ? atomicAdd(&n_heavy, 1) : 0;
            slot = __shfl(slot, 0, 64);
            if (slot < 64) {
                if (lane == 0) heavy[slot] = row;
                continue;
            }
        }
        float4 acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        csr_accumulate<NV, T>(x, col, wgt, e0, e1, 0, 1, cols, vec, lane, acc);
        csr_finish<NV, T>(acc, wgt, e1 > e0 ? 1.f / (float)(e1 - e0) : 0.f, gate, out, row, cols, vec, lane, -1, tee);
    }
    __syncthreads();
    const int nh = min(n_heavy, 64);  // (a workgroup walks <= rows/grid rows: far fewer than 64 heavy ones)
    for (int h = 0; h < nh; ++h) {
        // slots are claimed in arrival order: process them in ROW order so results do not depend on timing
        int row = 0x7fffffff;
        for (int q = 0; q < nh; ++q) {  // h-th smallest row id (nh is tiny)
            int below = 0;
            for (int r = 0; r < nh; ++r) below += heavy[r] < heavy[q];
            if (below == h) row = heavy[q];
        }
        const int e0 = rowptr[row], e1 = rowptr[row + 1];
        float4 acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        csr_accumulate<NV, T>(x, col, wgt, e0, e1, wave, WPB, cols, vec, lane, acc);
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(part + (wave - 1) * NV * 256 + (i * 64 + lane) * 4) = acc[i];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int wv = 0; wv < WPB - 1; ++wv)
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const float4 p = *reinterpret_cast<const float4*>(part + wv * NV * 256 + (i * 64 + lane) * 4);
                    acc[i].x += p.x; acc[i].y += p.y; acc[i].z += p.z; acc[i].w += p.w;
                }
            csr_finish<NV, T>(acc, wgt, 1.f / (float)(e1 - e0), gate, out, row, cols, vec, lane, -1, tee);
        }
        __syncthreads();
    }
}

// ---- bf16 rows of 1024 columns (rows1024.h): the banded MEAN gather of the forward pass ---------------------------------------
// RB rows per wave per sweep: their neighbour codes first (one byte each), then EVERY neighbour row of the sweep is requested
// before the first is used.  Coded rows add {i - 1, i, i + 1} in ascending order, general rows (0xFF: the LTA forecast nodes)
// walk their CSR entries in order -- the sums of csr_gather_kernel, bit for bit.  No listed heavy rows (host-checked).
template <int RB>
__global__ __launch_bounds__(256) void csr_gather_band_1k_kernel(const bf16_t* __restrict__ x, const int* __restrict__ rowptr,
                                                                 const int* __restrict__ col, const unsigned char* __restrict__ band,
                                                                 bf16_t* __restrict__ out, int rows, int skip_above) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    const int W = rw.step, rend = rw.end;
    for (int base = rw.first; base < rend; base += RB * W) {
        unsigned code[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) code[k] = base + k * W < rend ? band[base + k * W] : 0u;
        r1k::Raw nb[RB][3];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int row = base + k * W;
            if (row < rend && code[k] != 0xFFu) {
                const bf16_t* r0 = x + (long long)row * r1k::COLS;
                if (code[k] & 1u) nb[k][0] = r1k::ld_raw(r0 - r1k::COLS, lane);
                if (code[k] & 2u) nb[k][1] = r1k::ld_raw(r0, lane);
                if (code[k] & 4u) nb[k][2] = r1k::ld_raw(r0 + r1k::COLS, lane);
            }
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int row = base + k * W;
            if (row >= rend) break;
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            int cnt;
            if (code[k] != 0xFFu) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (code[k] & (1u << q)) {
                        float v[16];
                        r1k::unpack(nb[k][q], v);
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc[j] += 1.f * v[j];
                    }
                cnt = __popc(code[k] & 7u);
            } else {
                const int e0 = rowptr[row], e1 = rowptr[row + 1];
                cnt = e1 - e0;
                if (cnt > skip_above) continue;  // listed by the host: the split launches produce this row
                for (int e = e0; e < e1; e += 4) {  // 4 neighbour rows in flight, added in edge order
                    r1k::Raw t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (e + u < e1) t[u] = r1k::ld_raw(x + (long long)col[e + u] * r1k::COLS, lane);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (e + u < e1) {
                            float v[16];
                            r1k::unpack(t[u], v);
#pragma unroll
                            for (int j = 0; j < 16; ++j) acc[j] += 1.f * v[j];
                        }
                }
            }
            const float mean_w = cnt ? 1.f / (float)cnt : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] *= mean_w;
            r1k::st_raw(out + (long long)row * r1k::COLS, lane, r1k::pack(acc));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- bf16 rows of 1024 columns: the general (weighted / gated) gather at <= 96 registers -------------------------------------
// The backward pass's transposed gather (per-edge weights 1 / in-degree of the target, ReLU gate of the projection output) runs
// BESIDE grouped weight-gradient launches whose two workgroups per CU leave 96 registers per SIMD lane: a row kernel that needs
// more waits for a weight-gradient workgroup to retire (59.6 us instead of 16 for csr_gather_kernel<4> at 127 registers,
// profiles/r04_c3_replay_timeline.txt).  This kernel stays under that line without spilling: a row is held as packed bf16
// (rows1024.h) until it is added, at most four neighbour rows + the gate row are in flight per wave, the row's CSR bounds and
// entries are wave-uniform (scalar) fetches, and every wave owns whole rows (one row per wave per sweep, grid = rows / 4).
// Sums: edge order for every row that one wave sums (csr_gather_kernel's order for rows of <= HEAVY edges; rows of HEAVY + 1 ..
// skip_above edges are summed in edge order here and by four cooperating waves there: equal to rounding); the listed rows
// (> skip_above edges) are summed by one workgroup each, a quarter of the columns per wave, in edge order -- as there.
template <bool WGT, bool GATE>
__global__ __launch_bounds__(256, 5) void csr_gather_1k_kernel(const bf16_t* __restrict__ x, const int* __restrict__ rowptr,
                                                               const int* __restrict__ col, const float* __restrict__ wgt,
                                                               const bf16_t* __restrict__ gate, bf16_t* __restrict__ out, int rows,
                                                               int skip_above, const int* __restrict__ block_rows, int n_block_rows) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)blockIdx.x < n_block_rows) {  // a listed row: wave w sums columns [256 w, 256 w + 256), 4 per lane, 8 rows in flight
        const int row = block_rows[blockIdx.x];
        const int e0 = rowptr[row], e1 = rowptr[row + 1];
        const long long c = (long long)wave * 256 + lane * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int e = e0; e < e1; e += 8) {
            uint2 v[8];
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e + u < e1) {
                    v[u] = *reinterpret_cast<const uint2*>(x + (long long)col[e + u] * r1k::COLS + c);
                    w[u] = WGT ? wgt[e + u] : 1.f;
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e + u < e1) {
                    a0 += w[u] * __uint_as_float(v[u].x << 16); a1 += w[u] * __uint_as_float(v[u].x & 0xffff0000u);
                    a2 += w[u] * __uint_as_float(v[u].y << 16); a3 += w[u] * __uint_as_float(v[u].y & 0xffff0000u);
                }
        }
        if (!WGT) {
            const float mw = e1 > e0 ? 1.f / (float)(e1 - e0) : 0.f;
            a0 *= mw; a1 *= mw; a2 *= mw; a3 *= mw;
        }
        if (GATE) {
            const uint2 gq = *reinterpret_cast<const uint2*>(gate + (long long)row * r1k::COLS + c);
            a0 = __uint_as_float(gq.x << 16) > 0.f ? a0 : 0.f; a1 = __uint_as_float(gq.x & 0xffff0000u) > 0.f ? a1 : 0.f;
            a2 = __uint_as_float(gq.y << 16) > 0.f ? a2 : 0.f; a3 = __uint_as_float(gq.y & 0xffff0000u) > 0.f ? a3 : 0.f;
        }
        uint2 o;
        o.x = (unsigned)f2bf(a0) | ((unsigned)f2bf(a1) << 16);
        o.y = (unsigned)f2bf(a2) | ((unsigned)f2bf(a3) << 16);
        *reinterpret_cast<uint2*>(out + (long long)row * r1k::COLS + c) = o;
        return;
    }
    const int bid = blockIdx.x - n_block_rows, nblk = gridDim.x - n_block_rows;
    // (XCD-contiguous row ownership -- common.h -- when the listed rows in front keep blockIdx % 8 == bid % 8)
    const RowWalk rw = (n_block_rows & 7) == 0 ? row_walk(bid, nblk, 0, rows, wave, WPB) : RowWalk{bid * WPB + wave, rows, nblk * WPB};
    for (int row = rw.first; row < rw.end; row += rw.step) {  // (row is wave-uniform: scalar fetches below)
        const int e0 = rowptr[row], e1 = rowptr[row + 1];
        if (e1 - e0 > skip_above) continue;  // listed by the host
        r1k::Raw gt;
        if (GATE) gt = r1k::ld_raw(gate + (long long)row * r1k::COLS, lane);
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int e = e0; e < e1; e += 4) {
            r1k::Raw t[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u < e1) {
                    t[u] = r1k::ld_raw(x + (long long)col[e + u] * r1k::COLS, lane);
                    w[u] = WGT ? wgt[e + u] : 1.f;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u < e1) {
                    float v[16];
                    r1k::unpack(t[u], v);
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[j] += w[u] * v[j];
                }
        }
        if (!WGT) {
            const float mw = e1 > e0 ? 1.f / (float)(e1 - e0) : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] *= mw;
        }
        if (GATE) {
            float gv[16];
            r1k::unpack(gt, gv);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = gv[j] > 0.f ? acc[j] : 0.f;
        }
        r1k::st_raw(out + (long long)row * r1k::COLS, lane, r1k::pack(acc));
    }
}

// y = x + table[pos - pos_min] for bf16 rows of 1024 columns; positions outside the table are evaluated directly
template <int RB>
__global__ __launch_bounds__(256) void pe_add_table_1k_kernel(const bf16_t* __restrict__ x, const long long* __restrict__ pos,
                                                              const float* __restrict__ freq, const float* __restrict__ table,
                                                              long long pos_min, int n_pos, bf16_t* __restrict__ y, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    const int W = rw.step, rend = rw.end;
    for (int base = rw.first; base < rend; base += RB * W) {
        r1k::Raw raw[RB];
        long long pr[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k)
            if (base + k * W < rend) {
                raw[k] = r1k::ld_raw(x + (long long)(base + k * W) * r1k::COLS, lane);
                pr[k] = pos[base + k * W];
            }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int row = base + k * W;
            if (row >= rend) break;
            const long long ti = pr[k] - pos_min;
            const bool hit = ti >= 0 && ti < n_pos;  // (wave-uniform)
            float v[16], e[16];
            r1k::unpack(raw[k], v);
            if (hit) {
                r1k::ld_vec16(table + ti * r1k::COLS, lane, e);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int cc = r1k::col_of(lane, j);
                    const bool is_sin = cc < 512;
                    const float a = (float)pr[k] * freq[is_sin ? cc : cc - 512];
                    e[j] = is_sin ? sinf(a) : cosf(a);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] += e[j];
            r1k::st_raw(y + (long long)row * r1k::COLS, lane, r1k::pack(v));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- rows with hundreds of edges (the LTA fan-out node at T = 256 has out-degree 255) ------------------------------------
// One such row per sequence would keep ONE workgroup busy for ~100 us of dependent index -> row round trips while
// the rest of the chip idles.  The host lists them (data.build_csr: degree > VERY_HEAVY); each listed row is cut
// into CSR_CHUNKS edge ranges summed by separate workgroups (thread = 4 columns, the chunk's indices fetched up
// front, 8 neighbour rows in flight) into f32 partial rows, and a finish launch adds the partials in chunk order and
// applies weights / mean / gate.  No atomics: bitwise reproducible.
constexpr int VERY_HEAVY = 24, CSR_CHUNKS = 8;

template <typename T>
__global__ __launch_bounds__(256) void csr_heavy_partial_kernel(const T* __restrict__ x, const int* __restrict__ rowptr,
                                                                const int* __restrict__ col, const float* __restrict__ wgt,
                                                                const int* __restrict__ heavy, float* __restrict__ ws,
                                                                int cols) {
    const int h = blockIdx.x, ch = blockIdx.y;
    const int row = heavy[h];
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    const int per = (e1 - e0 + CSR_CHUNKS - 1) / CSR_CHUNKS;
    const int c_begin = e0 + ch * per, c_end = min(e1, c_begin + per);
    const bool vec = (cols & 3) == 0;
    float* wrow = ws + ((long long)h * CSR_CHUNKS + ch) * cols;
    for (int c0 = threadIdx.x * 4; c0 < cols; c0 += 1024) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = c_begin; e < c_end; e += 8) {
            const int cnt = min(8, c_end - e);  // (workgroup-uniform)
            int ci[8];
            float w[8];
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = u < cnt ? e + u : e;
                ci[u] = col[ee];
                w[u] = wgt ? wgt[ee] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld4(x + (long long)ci[u] * cols, c0, cols, vec);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < cnt) {
                    a.x += w[u] * v[u].x; a.y += w[u] * v[u].y; a.z += w[u] * v[u].z; a.w += w[u] * v[u].w;
                }
        }
        if (c0 + 4 <= cols) *reinterpret_cast<float4*>(wrow + c0) = a;
        else {
            const float t[4] = {a.x, a.y, a.z, a.w};
            for (int q = 0; q < 4 && c0 + q < cols; ++q) wrow[c0 + q] = t[q];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void csr_heavy_finish_kernel(const float* __restrict__ ws, const int* __restrict__ rowptr,
                                                               const float* __restrict__ wgt, const T* __restrict__ gate,
                                                               const int* __restrict__ heavy, T* __restrict__ out, int cols) {
    const int h = blockIdx.x;
    const int row = heavy[h];
    const float mean_w = 1.f / (float)(rowptr[row + 1] - rowptr[row]);
    for (int c = threadIdx.x; c < cols; c += 256) {
        float a = 0.f;
#pragma unroll
        for (int ch = 0; ch < CSR_CHUNKS; ++ch) a += ws[((long long)h * CSR_CHUNKS + ch) * cols + c];
        if (!wgt) a *= mean_w;
        if (gate && !(ld1t(gate + (long long)row * cols + c) > 0.f)) a = 0.f;
        st1t(out + (long long)row * cols + c, a);
    }
}

// ---- GraphONE gather-max ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_max_fwd_kernel(const T* __restrict__ f, const float* __restrict__ bank,
                                                             const long long* __restrict__ nn, T* __restrict__ m,
                                                             uint8_t* __restrict__ arg, int rows, int cols, int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (cols & 3) == 0;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        for (int c = lane * 4; c < cols; c += 256) {
            // message order of the reference: prototype edges first, the self loop appended last
            // (add_remaining_self_loops); first maximum wins on ties.
            float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            uint32_t a4 = 0;
            for (int j = 0; j <= k; ++j) {
                const float4 v = j < k ? ld4(bank + nn[(long long)row * k + j] * cols, c, cols, vec)
                                       : ld4(f + (long long)row * cols, c, cols, vec);
                if (v.x > best.x) { best.x = v.x; a4 = (a4 & ~0x000000ffu) | (uint32_t)j; }
                if (v.y > best.y) { best.y = v.y; a4 = (a4 & ~0x0000ff00u) | ((uint32_t)j << 8); }
                if (v.z > best.z) { best.z = v.z; a4 = (a4 & ~0x00ff0000u) | ((uint32_t)j << 16); }
                if (v.w > best.w) { best.w = v.w; a4 = (a4 & ~0xff000000u) | ((uint32_t)j << 24); }
            }
            st4(m + (long long)row * cols, c, cols, vec, best);
            uint8_t* ap = arg + (long long)row * cols + c;
            if (vec && c + 4 <= cols) *reinterpret_cast<uint32_t*>(ap) = a4;
            else
                for (int t = 0; t < 4 && c + t < cols; ++t) ap[t] = (a4 >> (8 * t)) & 0xff;
        }
    }
}

// The same op with every load of a row requested up front: the generic kernel walks k + 1 sources x cols / 256 column steps as
// a chain of dependent loads (one wave per row: ~15 us for 2048 rows whatever the bandwidth); here the k neighbour indices are
// read first and then U column steps x (K + 1) sources are in flight together.  Same comparisons in the same order (prototype
// edges first, self loop last, first maximum wins): bit-identical values and winners.  Rows of up to 4 GROUPS (tasks: own bank
// and neighbour lists, ``rows`` rows each, f / m / arg one block below the other) in one launch.
struct GatherMaxGroups {
    const float* bank[4];
    const long long* nn[4];
    int n_groups, rows;
};

template <typename T, int K, int U>
__global__ __launch_bounds__(256) void gather_max_fwd_u_kernel(const T* __restrict__ f, GatherMaxGroups gg, T* __restrict__ m,
                                                               uint8_t* __restrict__ arg, int rows_total, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows_total, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const int g = row / gg.rows, r = row - g * gg.rows;
        const float* __restrict__ bank = g == 0 ? gg.bank[0] : g == 1 ? gg.bank[1] : g == 2 ? gg.bank[2] : gg.bank[3];
        const long long* __restrict__ nn = (g == 0 ? gg.nn[0] : g == 1 ? gg.nn[1] : g == 2 ? gg.nn[2] : gg.nn[3]) + (long long)r * K;
        const float* src[K];
#pragma unroll
        for (int j = 0; j < K; ++j) src[j] = bank + nn[j] * cols;
        const T* self = f + (long long)row * cols;
        for (int c0 = lane * 4; c0 < cols; c0 += 256 * U) {
            float4 v[U][K + 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int j = 0; j < K; ++j) v[u][j] = ld4(src[j], c0 + 256 * u, cols, true);
                v[u][K] = ld4(self, c0 + 256 * u, cols, true);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                uint32_t a4 = 0;
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    const float4 x = v[u][j];
                    if (x.x > best.x) { best.x = x.x; a4 = (a4 & ~0x000000ffu) | (uint32_t)j; }
                    if (x.y > best.y) { best.y = x.y; a4 = (a4 & ~0x0000ff00u) | ((uint32_t)j << 8); }
                    if (x.z > best.z) { best.z = x.z; a4 = (a4 & ~0x00ff0000u) | ((uint32_t)j << 16); }
                    if (x.w > best.w) { best.w = x.w; a4 = (a4 & ~0xff000000u) | ((uint32_t)j << 24); }
                }
                const int c = c0 + 256 * u;
                st4(m + (long long)row * cols, c, cols, true, best);
                *reinterpret_cast<uint32_t*>(arg + (long long)row * cols + c) = a4;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_max_bwd_kernel(const T* __restrict__ dm, const uint8_t* __restrict__ arg,
                                                             T* __restrict__ df, long long n, int k, int accumulate) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
         i += (long long)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) {
            const float4 g = ld4(dm + i, 0, 4, true);
            const uint32_t a4 = *reinterpret_cast<const uint32_t*>(arg + i);
            float4 o;
            o.x = ((a4 >> 0) & 0xff) == (uint32_t)k ? g.x : 0.f;
            o.y = ((a4 >> 8) & 0xff) == (uint32_t)k ? g.y : 0.f;
            o.z = ((a4 >> 16) & 0xff) == (uint32_t)k ? g.z : 0.f;
            o.w = ((a4 >> 24) & 0xff) == (uint32_t)k ? g.w : 0.f;
            if (accumulate) {
                const float4 d = ld4(df + i, 0, 4, true);
                o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w;
            }
            st4(df + i, 0, 4, true, o);
        } else {
            for (long long j = i; j < n; ++j) {
                const float o = arg[j] == k ? ld1t(dm + j) : 0.f;
                st1t(df + j, accumulate ? ld1t(df + j) + o : o);
            }
        }
    }
}

// ---- per-sequence max pool ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void segmax_fwd_kernel(const T* __restrict__ x, const int* __restrict__ ptr,
                                                         T* __restrict__ out, int* __restrict__ arg, int n_seg,
                                                         int cols) {
    const int sg = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = ptr[sg], r1 = ptr[sg + 1];
    float best = 0.f;  // empty segment -> 0 (scatter 'amax' into zeros, include_self=False)
    int a = -1;
    for (int r = r0; r < r1; ++r) {
        const float v = ld1t(x + (long long)r * cols + c);
        if (a < 0 || v > best) {
            best = v;
            a = r;
        }
    }
    st1t(out + (long long)sg * cols + c, best);
    arg[(long long)sg * cols + c] = a;
}

// The same pool with the rows of a segment shared by four row lanes and four columns per thread: the thread-per-column walk
// above is one dependent 2-byte load per row -- 94 us for the 256-node sequences of BASELINE config 5 ([16 x 256, 1024] in,
// 16 rows out).  Workgroup = (segment, block of 256 columns); thread (cg = tid & 63, rl = tid >> 6) walks rows r0 + rl, r0 + rl
// + 4, ... eight at a time for columns 4 cg .. 4 cg + 3; the four lanes meet through LDS: the larger value wins, on equal
// values the SMALLER row -- the first occurrence, which is what the serial walk (strict >) keeps.  cols % 4 == 0.
// RL row lanes (waves) per workgroup: 4, or 16 when the launch has few segments (16 sequences of 256 nodes: 64 workgroups of 4
// waves walked 8 dependent rounds of loads each -- 44 us on the critical path of config 5 between the heads and backward).
// (up to four inputs over the same sequences in one launch -- blockIdx.z picks the input: the OSCC head pools the primary
//  features and one GraphONE output per auxiliary task, oscc.py:68,85)
template <typename T>
struct SegMaxSrcs {
    const T* x[4];
    T* out[4];
    int* arg[4];
};
template <typename T>
struct SegMaxGrads {
    const T* dout[4];
    const int* arg[4];
    T* dx[4];
};

template <typename T, int RL>
__global__ __launch_bounds__(64 * RL) void segmax_fwd_v4_kernel(SegMaxSrcs<T> src, const int* __restrict__ ptr, int n_seg, int cols) {
    __shared__ float sv[RL - 1][64][4];
    __shared__ int sa[RL - 1][64][4];
    const T* __restrict__ x = src.x[blockIdx.z];
    T* __restrict__ out = src.out[blockIdx.z];
    int* __restrict__ arg = src.arg[blockIdx.z];
    const int sg = blockIdx.y, cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + cg * 4;
    const bool live = c < cols;
    const int r0 = ptr[sg], r1 = ptr[sg + 1];
    float best[4] = {0.f, 0.f, 0.f, 0.f};  // empty segment -> 0 (scatter 'amax' into zeros, include_self=False)
    int a[4] = {-1, -1, -1, -1};
    if (live) {
        for (int r = r0 + rl; r < r1; r += 8 * RL) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + RL * u < r1) v[u] = ld4t(x + (long long)(r + RL * u) * cols, c, cols, true);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + RL * u < r1) {
                    const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (a[t] < 0 || e[t] > best[t]) {
                            best[t] = e[t];
                            a[t] = r + RL * u;
                        }
                }
        }
    }
    if (rl > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sv[rl - 1][cg][t] = best[t];
            sa[rl - 1][cg][t] = a[t];
        }
    }
    __syncthreads();
    if (rl == 0 && live) {
#pragma unroll
        for (int q = 0; q < RL - 1; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = sv[q][cg][t];
                const int av = sa[q][cg][t];
                if (av >= 0 && (a[t] < 0 || v > best[t] || (v == best[t] && av < a[t]))) {
                    best[t] = v;
                    a[t] = av;
                }
            }
        st4t(out + (long long)sg * cols, c, cols, true, make_float4(best[0], best[1], best[2], best[3]));
        *reinterpret_cast<int4*>(arg + (long long)sg * cols + c) = make_int4(a[0], a[1], a[2], a[3]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void segmax_bwd_kernel(SegMaxGrads<T> gr, const int* __restrict__ ptr, int n_seg, int cols) {
    const T* __restrict__ dout = gr.dout[blockIdx.z];
    const int* __restrict__ arg = gr.arg[blockIdx.z];
    T* __restrict__ dx = gr.dx[blockIdx.z];
    const int sg = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = ptr[sg], r1 = ptr[sg + 1];
    const int a = arg[(long long)sg * cols + c];
    const float g = ld1t(dout + (long long)sg * cols + c);
    for (int r = r0; r < r1; ++r) st1t(dx + (long long)r * cols + c, r == a ? g : 0.f);
}

// ---- cosine k-NN ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const T* __restrict__ x, float* __restrict__ inv, int rows,
                                                           int cols, int squared) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (cols & 3) == 0;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        float s = 0.f;
        for (int c = lane * 4; c < cols; c += 256) {
            const float4 v = ld4(x + (long long)row * cols, c, cols, vec);
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        s = wave_sum(s);
        if (lane == 0) inv[row] = squared ? s : 1.f / sqrtf(s);
    }
}

// the grouped search's three passes over the f32 feature rows as one: 1 / |row| (row_inv_norm_kernel's arithmetic: the same bits),
// the bf16 rounding (the activation-type copy the GraphONE stages read) and -- h16 != nullptr -- the IEEE-half rounding (the screen's
// operand on the f16 matrix instructions)
__global__ __launch_bounds__(256) void row_inv_norm_cast_kernel(const float* __restrict__ x, float* __restrict__ inv,
                                                                bf16_t* __restrict__ hi, unsigned short* __restrict__ h16, int rows,
                                                                int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const long long base = (long long)row * cols;
        float s = 0.f;
        for (int c = lane * 4; c < cols; c += 256) {  // (cols % 4 == 0: checked by the launcher)
            const float4 v = ld4(x + base, c, cols, true);
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            *reinterpret_cast<uint2*>(hi + base + c) = make_uint2((unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16),
                                                                  (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16));
            if (h16) {
                const _Float16 h[4] = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                *reinterpret_cast<uint2*>(h16 + base + c) = *reinterpret_cast<const uint2*>(h);
            }
        }
        s = wave_sum(s);
        if (lane == 0) inv[row] = 1.f / sqrtf(s);
    }
}

__global__ __launch_bounds__(256) void cos_dist_kernel(const float* __restrict__ dot, long long ldd,
                                                       const float* __restrict__ f_inv, const float* __restrict__ b_inv,
                                                       float* __restrict__ dist, int rows, int K) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (j < K) dist[(long long)r * K + j] = 1.f - dot[(long long)r * ldd + j] * f_inv[r] * b_inv[j];
}

// One wave per row, ONE pass over the row: every lane keeps the KM smallest (distance, index) pairs of the columns it
// visits (16-byte loads, 4 consecutive columns per lane and step) in a sorted register list -- almost every element
// fails the first comparison against the list's worst entry -- then k rounds of a wave-wide lexicographic arg-min
// over the list heads pick the row's k nearest in order (the winning lane pops its head).  Ties go to the smaller
// index, NaN distances are never selected, exactly as a full (distance, index) sort would order them.
// L2 = false: d = 1 - dot * f_inv[n] * b_inv[j] (cos_dissimilarity, graphONE.py:148-151);
// L2 = true : d = sqrt(max(|f_n|^2 + |p_j|^2 - 2 dot, 0)) / 4096 (cdist / 4096, graphONE.py:126-127,144-145) with the
//             squared norms passed in f_inv / b_inv.
template <int KM, bool L2>
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ dot, long long ldd,
                                                   const float* __restrict__ f_inv, const float* __restrict__ b_inv,
                                                   long long* __restrict__ nn, int rows, int K, int k, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const float* dr = dot + (long long)row * ldd;
        const float fi = f_inv[row];
        float lv[KM];
        int li[KM];
#pragma unroll
        for (int s = 0; s < KM; ++s) {
            lv[s] = INFINITY;
            li[s] = 0x7fffffff;
        }
        // four steps of 256 columns are REQUESTED before the first is examined (a row is 16 dependent round trips otherwise:
        // 26-51 us for 2048 rows of 4096 against 8 us of bytes); examined in column order, as before
        for (int base0 = lane * 4; base0 < K; base0 += 1024) {
          float dq[4][4], bq[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int base = base0 + 256 * u;
            if (vec && base + 4 <= K) {
                const float4 a = *reinterpret_cast<const float4*>(dr + base);
                const float4 b = *reinterpret_cast<const float4*>(b_inv + base);
                dq[u][0] = a.x; dq[u][1] = a.y; dq[u][2] = a.z; dq[u][3] = a.w;
                bq[u][0] = b.x; bq[u][1] = b.y; bq[u][2] = b.z; bq[u][3] = b.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dq[u][t] = base + t < K ? dr[base + t] : 0.f;
                    bq[u][t] = base + t < K ? b_inv[base + t] : 0.f;
                }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int base = base0 + 256 * u;
            const float (&dv)[4] = dq[u];
            const float (&bv4)[4] = bq[u];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = base + t;
                float d = L2 ? sqrtf(fmaxf(fi + bv4[t] - 2.f * dv[t], 0.f)) * (1.f / 4096.f) : 1.f - dv[t] * fi * bv4[t];
                int dj = j;
                if (j < K && (d < lv[KM - 1] || (d == lv[KM - 1] && dj < li[KM - 1]))) {
#pragma unroll
                    for (int s = 0; s < KM; ++s) {  // sorted insert: carry the larger pair down the list
                        const bool before = d < lv[s] || (d == lv[s] && dj < li[s]);
                        const float tv = lv[s];
                        const int ti = li[s];
                        lv[s] = before ? d : tv;
                        li[s] = before ? dj : ti;
                        d = before ? tv : d;
                        dj = before ? ti : dj;
                    }
                }
            }
          }
        }
        for (int sel = 0; sel < k; ++sel) {
            float bv = lv[0];
            int bi = li[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov < bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (lane == 0) nn[(long long)row * k + sel] = bi == 0x7fffffff ? 0 : bi;
            if (li[0] == bi && bi != 0x7fffffff) {  // column indices are unique: exactly one lane owns the winner
#pragma unroll
                for (int s = 0; s + 1 < KM; ++s) {
                    lv[s] = lv[s + 1];
                    li[s] = li[s + 1];
                }
                lv[KM - 1] = INFINITY;
                li[KM - 1] = 0x7fffffff;
            }
        }
    }
}

// ---- nearest prototypes from ONE bf16 product: proven window + exact re-rank -------------------------------------------------
// The search ranks cos(f_n, p_j) (reference models/graphONE/graphONE.py:119-141,148-151: argsort of 1 - cos).  ``dot1`` is the
// bf16-MFMA product of hi(f) and hi(P), hi(x) = the bf16 nearest to x: one third of the matrix work of the three-product (f32-grade)
// contraction.  Its error is bounded per row: dot - dot1 = f_lo . P_hi + f_hi . P_lo + f_lo . P_lo with x_lo = x - hi(x), so by
// Cauchy-Schwarz |cos - cos1| <= rf (1 + rb) + (1 + rf) rb + rf rb with rf = |f_lo| / |f| (computed here from the row itself) and
// rb = max_j |p_lo,j| / |p_j| (egk_bf16_residual_ratio, once per bank), plus the f32 accumulation of the H-term product
// (<= H 2^-24 relative to |f| |p|: 6.1e-5 at H = 1024) -- E below.  With D_k the k-th smallest approximate distance of the row, a
// prototype whose approximate distance exceeds D_k + 2 E cannot be among the k nearest: its true distance exceeds D_k + E, and k
// prototypes have true distances <= D_k + E.  Every other prototype (a handful: profiles/r05_window_search.txt) is a CANDIDATE and gets
// its EXACT product -- f32 operands, double accumulation, one wave per dot -- and the f32 key of topk_kernel; the k nearest candidates
// in (key, index) order are the result.  No cap on the candidates: a row with many of them is slow, never wrong.
__device__ __forceinline__ float bf16_rne_f32(float x) {
    const unsigned u = __float_as_uint(x);
    return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);  // (finite inputs: features / prototypes)
}

// (the f16 screen: the IEEE-half nearest to x -- inf beyond 65504, which makes the residual ratio, the window and with it the
//  candidate list unbounded: slow, never wrong)
__device__ __forceinline__ float f16_rne_f32(float x) { return (float)(_Float16)x; }
template <bool F16>
__device__ __forceinline__ float round16(float x) { return F16 ? f16_rne_f32(x) : bf16_rne_f32(x); }

template <bool F16>
__global__ __launch_bounds__(256) void bf16_residual_ratio_kernel(const float* __restrict__ x, long long ld, float* __restrict__ r,
                                                                  int rows, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        float s2 = 0.f, r2 = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float v = x[(long long)row * ld + c], d = v - round16<F16>(v);
            s2 += v * v;
            r2 += d * d;
        }
        s2 = wave_sum(s2);
        r2 = wave_sum(r2);
        if (lane == 0) r[row] = s2 > 0.f ? sqrtf(r2 / s2) : 0.f;
    }
}
__global__ __launch_bounds__(256) void cast_f16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long n) {
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            const _Float16 h[4] = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            *reinterpret_cast<uint2*>(y + i) = *reinterpret_cast<const uint2*>(h);
        } else {
            for (long long j = i; j < n; ++j) {
                const _Float16 h = (_Float16)x[j];
                y[j] = *reinterpret_cast<const unsigned short*>(&h);
            }
        }
    }
}
__global__ __launch_bounds__(256) void max_of_kernel(const float* __restrict__ r, float* __restrict__ out, int n) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, r[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Several searches in one launch: rows [g * rows_per_group, (g + 1) * rows_per_group) of dot1 / f / f_inv / nn belong to bank g (the
// auxiliary tasks of one EgoPack batch: their feature rows are consecutive blocks of one buffer).
struct TopkWindowBanks {
    const float* bank[8];
    const float* b_inv[8];
    const float* rb_max[8];
    int rows_per_group;
    int screen_f16;  // dot1 came from IEEE-half roundings of the operands (else bf16): the row's own residual is taken accordingly
};

template <int KM>
__global__ __launch_bounds__(256, 3) void topk_window_kernel(const float* __restrict__ dot1, long long ldd, const float* __restrict__ f,
                                                          long long ldf, TopkWindowBanks tb, long long ldb,
                                                          const float* __restrict__ f_inv, long long* __restrict__ nn,
                                                          int* __restrict__ cand, int rows, int K, int H, int k, int vec) {
    constexpr int CAND_CAP = 512;  // candidates listed per row (real prototype banks: 30-90 per row, up to ~200; beyond: the rescan)
    __shared__ int s_cand[WPB][CAND_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const int grp = row / tb.rows_per_group;  // (wave-uniform)
        const float* __restrict__ bank = tb.bank[grp];
        const float* __restrict__ b_inv = tb.b_inv[grp];
        const float rb = tb.rb_max[grp][0];
        const float* dr = dot1 + (long long)row * ldd;
        const float* fr = f + (long long)row * ldf;
        const float fi = f_inv[row];
        // the row's own rounding residual (and the row itself in registers when it fits: H <= 1024, whole float4 groups)
        const bool in_regs = (H <= 1024) && ((H & 3) == 0) && ((ldf & 3) == 0);
        float4 fv[4];
        float s2 = 0.f, r2 = 0.f;
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = lane * 4 + 256 * u;
                fv[u] = c < H ? *reinterpret_cast<const float4*>(fr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float e[4] = {fv[u].x, fv[u].y, fv[u].z, fv[u].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float d = e[t] - (tb.screen_f16 ? f16_rne_f32(e[t]) : bf16_rne_f32(e[t]));
                    s2 += e[t] * e[t];
                    r2 += d * d;
                }
            }
        } else {
            for (int c = lane; c < H; c += 64) {
                const float v = fr[c], d = v - (tb.screen_f16 ? f16_rne_f32(v) : bf16_rne_f32(v));
                s2 += v * v;
                r2 += d * d;
            }
        }
        s2 = wave_sum(s2);
        r2 = wave_sum(r2);
        const float rf = s2 > 0.f ? sqrtf(r2 / s2) : 0.f;
        const float E = 1.01f * (rf + rb) + 3.f * rf * rb + (float)H * 6.0e-8f + 2.0e-6f;

        // ---- pass 1: an upper bound U of the k-th smallest APPROXIMATE distance: the k-th smallest of the 64 lanes' minima (k distinct
        // prototypes lie at or below it, so nothing beyond U + 2 E can be among the k nearest; U IS the k-th smallest whenever the k
        // nearest fall into k different lanes -- 91 % of the rows at k = 4 -- and the next few otherwise: a candidate more) ---------
        float lmin = INFINITY;
        for (int base0 = lane * 4; base0 < K; base0 += 1024) {
            float dq[4][4], bq[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int base = base0 + 256 * u;
                if (vec && base + 4 <= K) {
                    const float4 a = *reinterpret_cast<const float4*>(dr + base);
                    const float4 b = *reinterpret_cast<const float4*>(b_inv + base);
                    dq[u][0] = a.x; dq[u][1] = a.y; dq[u][2] = a.z; dq[u][3] = a.w;
                    bq[u][0] = b.x; bq[u][1] = b.y; bq[u][2] = b.z; bq[u][3] = b.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        dq[u][t] = base + t < K ? dr[base + t] : 0.f;
                        bq[u][t] = base + t < K ? b_inv[base + t] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float d = 1.f - dq[u][t] * fi * bq[u][t];
                    if (base0 + 256 * u + t < K) lmin = fminf(lmin, d);
                }
        }
        float Dk = INFINITY;
        for (int sel = 0; sel < k; ++sel) {
            float bv = lmin;
            int bi = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov < bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            Dk = bv;
            if (lane == bi) lmin = INFINITY;
        }
        const float T = Dk + 2.f * E;  // (fewer than k lanes with a finite minimum: Dk = inf, every finite distance is a candidate)

        // ---- pass 2a: the candidates' indices, listed per wave (LDS; more than CAND_CAP of them: the rescan below) -----------------
        volatile int* cl = s_cand[wave];
        int n_cand = 0;
        for (int base0 = 0; base0 < K; base0 += 1024) {
            float dq[4][4], bq[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int base = base0 + lane * 4 + 256 * u;
                if (vec && base + 4 <= K) {
                    const float4 a = *reinterpret_cast<const float4*>(dr + base);
                    const float4 b = *reinterpret_cast<const float4*>(b_inv + base);
                    dq[u][0] = a.x; dq[u][1] = a.y; dq[u][2] = a.z; dq[u][3] = a.w;
                    bq[u][0] = b.x; bq[u][1] = b.y; bq[u][2] = b.z; bq[u][3] = b.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        dq[u][t] = base + t < K ? dr[base + t] : 0.f;
                        bq[u][t] = base + t < K ? b_inv[base + t] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int j = base0 + lane * 4 + 256 * u + t;
                    const float d = 1.f - dq[u][t] * fi * bq[u][t];
                    const bool pred = j < K && !(d > T);  // (a NaN screen value -- an operand beyond the half range -- is a candidate)
                    const unsigned long long mask = __ballot(pred);
                    if (mask) {  // (wave-uniform)
                        const int slot = n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (pred && slot < CAND_CAP) cl[slot] = j;
                        n_cand += __popcll(mask);
                    }
                }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- pass 2b: exact distances of the candidates; the k nearest of them in (key, index) order (wave-uniform list) -------------
        float ev[KM];
        int ei[KM];
#pragma unroll
        for (int s = 0; s < KM; ++s) {
            ev[s] = INFINITY;
            ei[s] = 0x7fffffff;
        }
        auto keep = [&](float de, int dj) {
            if (de < ev[KM - 1] || (de == ev[KM - 1] && dj < ei[KM - 1])) {
#pragma unroll
                for (int s = 0; s < KM; ++s) {
                    const bool before = de < ev[s] || (de == ev[s] && dj < ei[s]);
                    const float tv = ev[s];
                    const int ti = ei[s];
                    ev[s] = before ? de : tv;
                    ei[s] = before ? dj : ti;
                    de = before ? tv : de;
                    dj = before ? ti : dj;
                }
            }
        };
        auto one = [&](int jj) {  // one candidate, any shape
            const float* pr = bank + (long long)jj * ldb;
            double acc = 0.0;
            for (int c = lane; c < H; c += 64) acc += (double)fr[c] * (double)pr[c];
            acc = wave_sum(acc);
            keep(1.f - (float)acc * fi * b_inv[jj], jj);
        };
        if (n_cand > CAND_CAP) {
            // a crowded row: every candidate in scan order, one at a time (slow, never wrong)
            for (int j0 = 0; j0 < K; j0 += 64) {
                const int j = j0 + lane;
                const float d = j < K ? 1.f - dr[j] * fi * b_inv[j] : INFINITY;
                unsigned long long mask = __ballot(j < K && !(d > T));
                while (mask) {
                    const int b = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    one(j0 + b);
                }
            }
        } else if (!(in_regs && (ldb & 3) == 0)) {
            for (int c = 0; c < n_cand; ++c) one(__builtin_amdgcn_readfirstlane(cl[c]));
        } else {
            // four candidates at a time: their rows in TWO rounds of loads (32 registers in flight), the four lane-partial sums folded
            // TOGETHER -- each of the first two halving steps of the wave hands half of the sums to the partner lane, so four
            // totals cost 7 exchanges instead of 24 -- candidate i's total ends in lanes 16 i .. 16 i + 15
            for (int c0 = 0; c0 < n_cand; c0 += 4) {
                int cj[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) cj[i] = c0 + i < n_cand ? __builtin_amdgcn_readfirstlane(cl[c0 + i]) : -1;
                double acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = 0.0;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float4 pv[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2) {
                            // unconditional loads (a guarded load is a branch and a wait of its own): an empty slot reads row 0 and
                            // is dropped below; columns beyond H read the row's last four, against zeros in fv
                            const int c = min(lane * 4 + 256 * (2 * half + q2), H - 4);
                            pv[i][q2] = *reinterpret_cast<const float4*>(bank + (long long)max(cj[i], 0) * ldb + c);
                        }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_barrier(0);  // (one candidate's widened elements live at a time)
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2) {
                            const float4 a = fv[2 * half + q2], b = pv[i][q2];
                            acc[i] += (double)a.x * (double)b.x + (double)a.y * (double)b.y + (double)a.z * (double)b.z +
                                      (double)a.w * (double)b.w;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                double a2[2], a1;
                {
                    const bool hi = lane & 32;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const double recv = __shfl_xor(hi ? acc[i] : acc[i + 2], 32, 64);
                        a2[i] = (hi ? acc[i + 2] : acc[i]) + recv;
                    }
                }
                {
                    const bool hi = lane & 16;
                    const double recv = __shfl_xor(hi ? a2[0] : a2[1], 16, 64);
                    a1 = (hi ? a2[1] : a2[0]) + recv;
                }
                a1 += __shfl_xor(a1, 8, 64);
                a1 += __shfl_xor(a1, 4, 64);
                a1 += __shfl_xor(a1, 2, 64);
                a1 += __shfl_xor(a1, 1, 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double tot = __shfl(a1, 16 * i, 64);
                    if (cj[i] >= 0) keep(1.f - (float)tot * fi * b_inv[cj[i]], cj[i]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            for (int s = 0; s < k; ++s) nn[(long long)row * k + s] = (s < KM && ei[s] != 0x7fffffff) ? ei[s] : 0;
            if (cand) cand[row] = n_cand;
        }
    }
}

// ---- prototype bank accumulation (fp64 bank, no atomics) -----------------------------------------------------------
// reference graphone.py:53: bank += scatter(task_feat, labels, dim_size, reduce="sum") -- the per-batch scatter sums the
// rows of one label in the feature type (fp32) in node order, then the [size, H] result is added to the fp64 bank.  Here
// the labelled rows arrive grouped by label (``order`` = node ids stably sorted by label, ``seg_ptr`` the group
// boundaries, ``seg_label`` the label of each group: integer work done by the caller) and ONE wave owns a group: it sums
// the group's rows in fp32 in node order -- the summation order of the reference's CPU scatter_add_ -- and adds the sum
// to its bank row in fp64.  Groups of one call have distinct labels, so no two waves touch the same bank row: no atomics,
// bitwise reproducible.  count[label] += group size.
template <typename T>
__global__ __launch_bounds__(256) void segment_sum_f64_kernel(const T* __restrict__ x, const int* __restrict__ order,
                                                              const int* __restrict__ seg_ptr,
                                                              const long long* __restrict__ seg_label, double* __restrict__ bank,
                                                              long long* __restrict__ count, int n_seg, int cols,
                                                              long long n_labels) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (cols & 3) == 0;
    for (int sg = blockIdx.x * WPB + wave; sg < n_seg; sg += gridDim.x * WPB) {
        const long long lb = seg_label[sg];
        if (lb < 0 || lb >= n_labels) continue;
        const int r0 = seg_ptr[sg], r1 = seg_ptr[sg + 1];
        if (lane == 0 && count) count[lb] += r1 - r0;
        for (int c = lane * 4; c < cols; c += 256) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            int r = r0;
            for (; r + 4 <= r1; r += 4) {  // 4 rows in flight, added in node order
                const int i0 = order[r], i1 = order[r + 1], i2 = order[r + 2], i3 = order[r + 3];
                const float4 v0 = ld4(x + (long long)i0 * cols, c, cols, vec), v1 = ld4(x + (long long)i1 * cols, c, cols, vec);
                const float4 v2 = ld4(x + (long long)i2 * cols, c, cols, vec), v3 = ld4(x + (long long)i3 * cols, c, cols, vec);
                a.x = ((a.x + v0.x) + v1.x) + v2.x + v3.x; a.y = ((a.y + v0.y) + v1.y) + v2.y + v3.y;
                a.z = ((a.z + v0.z) + v1.z) + v2.z + v3.z; a.w = ((a.w + v0.w) + v1.w) + v2.w + v3.w;
            }
            for (; r < r1; ++r) {
                const float4 v = ld4(x + (long long)order[r] * cols, c, cols, vec);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            double* b = bank + lb * cols + c;
            if (c + 0 < cols) b[0] += (double)a.x;
            if (c + 1 < cols) b[1] += (double)a.y;
            if (c + 2 < cols) b[2] += (double)a.z;
            if (c + 3 < cols) b[3] += (double)a.w;
        }
    }
}

// ---- GraphONE gather-max: gradient of the (trainable) prototype rows ---------------------------------------------------
// dbank[p, c] += sum over the edges e = (n, j) with nn[n, j] == p of (arg[n, c] == j ? dm[n, c] : 0): a gather over the
// edge list grouped by prototype (t_rowptr [K+1], t_edge = n*k + j ascending inside a group), one wave per prototype
// row, fixed summation order -- no atomics.
template <typename T>
__global__ __launch_bounds__(256) void gather_max_bank_grad_kernel(const T* __restrict__ dm, const uint8_t* __restrict__ arg,
                                                                   const int* __restrict__ t_rowptr,
                                                                   const int* __restrict__ t_edge, float* __restrict__ dbank,
                                                                   int K, int cols, int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (cols & 3) == 0;
    for (int p = blockIdx.x * WPB + wave; p < K; p += gridDim.x * WPB) {
        const int e0 = t_rowptr[p], e1 = t_rowptr[p + 1];
        if (e0 == e1) continue;
        for (int c = lane * 4; c < cols; c += 256) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = e0; e < e1; ++e) {
                const int ed = t_edge[e], n = ed / k;
                const uint32_t j = (uint32_t)(ed - n * k);
                const float4 g = ld4(dm + (long long)n * cols, c, cols, vec);
                const uint8_t* ap = arg + (long long)n * cols + c;
                if (c + 0 < cols && ap[0] == j) a.x += g.x;
                if (c + 1 < cols && ap[1] == j) a.y += g.y;
                if (c + 2 < cols && ap[2] == j) a.z += g.z;
                if (c + 3 < cols && ap[3] == j) a.w += g.w;
            }
            float* d = dbank + (long long)p * cols + c;
            if (c + 0 < cols) d[0] += a.x;
            if (c + 1 < cols) d[1] += a.y;
            if (c + 2 < cols) d[2] += a.z;
            if (c + 3 < cols) d[3] += a.w;
        }
    }
}

// element-type conversion f32 <-> bf16 (inputs of the bf16 pipeline, f32 views for the exact k-NN path)
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long long n, int vec) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
         i += (long long)gridDim.x * blockDim.x * 4) {
        if (vec && i + 4 <= n) st4t(dst + i, 0, 4, true, ld4t(src + i, 0, 4, true));
        else
            for (long long j = i; j < n && j < i + 4; ++j) st1t(dst + j, ld1t(src + j));
    }
}

// x = hi + lo, hi = bf16(x), lo = bf16(x - hi): the bf16 operand pair of a three-product (f32-grade) contraction.
// 4 consecutive elements per thread (16-byte load, 8-byte stores) when rows and pointers allow, else element by element.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, long long lds_, bf16_t* __restrict__ hi,
                                                         bf16_t* __restrict__ lo, long long ldo, long long rows, long long cols, int vec) {
    const long long per_row = (cols + 3) / 4, total = rows * per_row;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const long long r = q / per_row, c = (q - r * per_row) * 4;
        const float* p = src + r * lds_ + c;
        float x[4];
        const bool full = c + 4 <= cols;
        if (vec && full) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = c + t < cols ? p[t] : 0.f;
        }
        bf16_t h[4], l[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            h[t] = f2bf(x[t]);
            l[t] = f2bf(x[t] - bf2f(h[t]));
        }
        if (vec && full) {
            if (hi) *reinterpret_cast<uint2*>(hi + r * ldo + c) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
            *reinterpret_cast<uint2*>(lo + r * ldo + c) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
        } else {
            for (int t = 0; t < 4 && c + t < cols; ++t) {
                if (hi) hi[r * ldo + c + t] = h[t];
                lo[r * ldo + c + t] = l[t];
            }
        }
    }
}

// row-strided conversion: dst[r, c] = src[r, c] for c < cols (different leading dimensions: builds the 16-byte
// aligned operand copy of a [rows, cols] gradient whose width is not a multiple of 8)
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_rows_kernel(const S* __restrict__ src, long long lds_, D* __restrict__ dst,
                                                        long long ldd, int rows, int cols, int zero_cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        for (int c = lane; c < cols; c += 64) st1t(dst + (long long)row * ldd + c, ld1t(src + (long long)row * lds_ + c));
        for (int c = cols + lane; c < zero_cols; c += 64) st1t(dst + (long long)row * ldd + c, 0.f);
    }
}

// ---- resident feature store: out[i, :] = table[idx[i], :] (idx < 0 -> zeros), any f32 / bf16 pair ---------------------
// one wave per output row, 16 bytes of the source row per lane and step; rows are 3-6 KB (1536 features), so every
// access is a full-width coalesced instruction and the kernel streams at the rate the gathered rows arrive from HBM.
template <typename S, typename D>
__global__ __launch_bounds__(256) void gather_rows_kernel(const S* __restrict__ table, long long ld,
                                                          const long long* __restrict__ idx, D* __restrict__ out,
                                                          long long n, int cols, long long table_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int V = 16 / sizeof(S);  // source elements per 16 bytes
    const bool vec = (cols % V == 0) && (ld % V == 0);
    // (XCD x writes a contiguous eighth of the output rows -- the rows the first contraction's tiles of XCD x read: common.h)
    const RowWalk rw = n < (1ll << 30) ? row_walk(blockIdx.x, gridDim.x, 0, (int)n, wave, WPB) : RowWalk{0, 0, 1};
    const long long first = n < (1ll << 30) ? rw.first : (long long)blockIdx.x * WPB + wave;
    const long long end = n < (1ll << 30) ? rw.end : n, step = n < (1ll << 30) ? rw.step : (long long)gridDim.x * WPB;
    for (long long row = first; row < end; row += step) {
        const long long src = idx[row];
        D* o = out + row * cols;
        if (src < 0 || src >= table_rows) {
            for (int c = lane; c < cols; c += 64) st1t(o + c, 0.f);
            continue;
        }
        const S* t = table + src * ld;
        if (vec) {
            for (int c = lane * V; c < cols; c += 64 * V) {
                if constexpr (sizeof(S) == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(t + c);
                    st4t(o, c, cols, true, v);
                } else {
                    const uint4 v = *reinterpret_cast<const uint4*>(t + c);
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
                    if constexpr (sizeof(D) == 2) {
                        *reinterpret_cast<uint4*>(o + c) = v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            o[c + 2 * q] = __uint_as_float(w[q] << 16);
                            o[c + 2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
                        }
                    }
                }
            }
        } else {
            for (int c = lane; c < cols; c += 64) st1t(o + c, ld1t(t + c));
        }
    }
}

// out[i, :] = table[lo[i], :] where lo[i] == hi[i] (a copy), else (float)((1 - w[i]) * table[lo[i], :] + w[i] * table[hi[i], :])
// evaluated in double with separately rounded products and sum -- numpy's
//     (1 - frac)[:, None] * low + frac[:, None] * high  ->  torch.from_numpy(...).float()
// of the PNR sampling (data/ego4d_oscc.py:258-275) bit for bit; an index < 0 stands for an all-zero row.
template <typename S, typename D>
__global__ __launch_bounds__(256) void gather_lerp_rows_kernel(const S* __restrict__ table, long long ld,
                                                               const long long* __restrict__ lo, const long long* __restrict__ hi,
                                                               const double* __restrict__ w, D* __restrict__ out, long long n,
                                                               int cols, long long table_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * WPB + wave; row < n; row += (long long)gridDim.x * WPB) {
        const long long a = lo[row], b = hi[row];
        const bool va = a >= 0 && a < table_rows, vb = b >= 0 && b < table_rows;
        const S* ta = table + (va ? a : 0) * ld;
        const S* tb = table + (vb ? b : 0) * ld;
        D* o = out + row * cols;
        if (a == b) {
            for (int c = lane; c < cols; c += 64) st1t(o + c, va ? ld1t(ta + c) : 0.f);
        } else {
            const double wb = w[row], wa = 1.0 - wb;
            for (int c = lane; c < cols; c += 64) {
                const double x = va ? (double)ld1t(ta + c) : 0.0, y = vb ? (double)ld1t(tb + c) : 0.0;
                st1t(o + c, (float)__dadd_rn(__dmul_rn(wa, x), __dmul_rn(wb, y)));
            }
        }
    }
}

static inline int row_grid(int rows) {
    int g = cdiv(rows, WPB);
    return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}
int g_graph_rows_v2 = 1;  // development knob (egk_tune 3, set from norm_ops.hip): the rows1024.h kernels of this file
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int g_gather_max_up_front = 1;  // development knob (egk_tune 7): 0 = the generic kernel everywhere

// Whether the loads-up-front kernel takes this shape; launches it when it does.
template <typename T>
static bool gather_max_up_front(hipStream_t s, const T* f, const GatherMaxGroups& gg, T* m, uint8_t* arg, int cols, int k) {
    const int steps = cols / 256;
    const uintptr_t amask = 4 * sizeof(T) - 1;  // (four elements per lane and access)
    if (!g_gather_max_up_front || cols % 256 || (reinterpret_cast<uintptr_t>(m) & amask) || (reinterpret_cast<uintptr_t>(f) & amask) ||
        (reinterpret_cast<uintptr_t>(arg) & 3))
        return false;
    for (int g = 0; g < gg.n_groups; ++g)
        if (!al16(gg.bank[g])) return false;
    const int total = gg.n_groups * gg.rows;
    if (k == 4 && steps % 4 == 0)
        hipLaunchKernelGGL((gather_max_fwd_u_kernel<T, 4, 4>), dim3(row_grid(total)), dim3(256), 0, s, f, gg, m, arg, total, cols);
    else if (k == 4 && steps % 2 == 0)
        hipLaunchKernelGGL((gather_max_fwd_u_kernel<T, 4, 2>), dim3(row_grid(total)), dim3(256), 0, s, f, gg, m, arg, total, cols);
    else if (k == 8 && steps % 2 == 0)
        hipLaunchKernelGGL((gather_max_fwd_u_kernel<T, 8, 2>), dim3(row_grid(total)), dim3(256), 0, s, f, gg, m, arg, total, cols);
    else if (k == 4)
        hipLaunchKernelGGL((gather_max_fwd_u_kernel<T, 4, 1>), dim3(row_grid(total)), dim3(256), 0, s, f, gg, m, arg, total, cols);
    else if (k == 8)
        hipLaunchKernelGGL((gather_max_fwd_u_kernel<T, 8, 1>), dim3(row_grid(total)), dim3(256), 0, s, f, gg, m, arg, total, cols);
    else
        return false;
    return true;
}


}  // namespace egk

using namespace egk;

extern "C" {

int egk_cast(egk_stream_t stream, const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n) {
    EGK_REQUIRE(src && dst, "egk_cast: null pointer");
    EGK_REQUIRE(src_dtype != dst_dtype, "egk_cast: source and destination types are equal");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_CAST, s, 0, 6.0 * n);
    const int vec = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    long long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
    if (src_dtype == EGK_F32 && dst_dtype == EGK_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, (long long)n, vec);
    else if (src_dtype == EGK_BF16 && dst_dtype == EGK_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, (long long)n, vec);
    else {
        set_error("egk_cast: unsupported type pair %d -> %d", src_dtype, dst_dtype);
        return EGK_EUNSUPPORTED;
    }
    return check_launch("egk_cast");
}

int egk_split_bf16(egk_stream_t stream, const float* src, int64_t ld_src, void* hi, void* lo, int64_t ld_out, int64_t rows,
                   int64_t cols) {
    EGK_REQUIRE(src && lo, "egk_split_bf16: null pointer");
    EGK_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols && ld_out >= cols, "egk_split_bf16: bad shape / leading dimension");
    if (rows == 0 || cols == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_CAST, s, 0, (hi ? 8.0 : 6.0) * rows * cols);
    const int vec = (((uintptr_t)src & 15) == 0) && (ld_src % 4 == 0) && ((((uintptr_t)lo | (uintptr_t)hi) & 7) == 0) && (ld_out % 4 == 0);
    long long blocks = (rows * ((cols + 3) / 4) + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, (long long)ld_src, (bf16_t*)hi, (bf16_t*)lo,
                       (long long)ld_out, (long long)rows, (long long)cols, vec);
    return check_launch("egk_split_bf16");
}

int egk_cast_rows(egk_stream_t stream, const void* src, int32_t src_dtype, int64_t ld_src, void* dst, int32_t dst_dtype,
                  int64_t ld_dst, int32_t rows, int32_t cols, int32_t zero_cols) {
    EGK_REQUIRE(src && dst, "egk_cast_rows: null pointer");
    EGK_REQUIRE(zero_cols <= ld_dst, "egk_cast_rows: zero_cols beyond the destination row");
    if (rows == 0 || cols == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_CAST, s, 0, 6.0 * rows * cols);
    const dim3 grid(row_grid(rows)), block(256);
#define EGK_CR(S, D) hipLaunchKernelGGL((cast_rows_kernel<S, D>), grid, block, 0, s, (const S*)src, (long long)ld_src, (D*)dst, (long long)ld_dst, rows, cols, zero_cols)
    if (src_dtype == EGK_F32 && dst_dtype == EGK_BF16) EGK_CR(float, bf16_t);
    else if (src_dtype == EGK_BF16 && dst_dtype == EGK_F32) EGK_CR(bf16_t, float);
    else if (src_dtype == EGK_F32 && dst_dtype == EGK_F32) EGK_CR(float, float);
    else if (src_dtype == EGK_BF16 && dst_dtype == EGK_BF16) EGK_CR(bf16_t, bf16_t);
    else {
        set_error("egk_cast_rows: unknown element type");
        return EGK_EINVAL;
    }
#undef EGK_CR
    return check_launch("egk_cast_rows");
}

int egk_gather_rows(egk_stream_t stream, const void* table, int32_t table_dtype, int64_t ld, int64_t table_rows,
                    const int64_t* idx, void* out, int32_t out_dtype, int64_t n, int32_t cols) {
    EGK_REQUIRE(table && idx && out, "egk_gather_rows: null pointer");
    EGK_REQUIRE(cols >= 1 && ld >= cols, "egk_gather_rows: bad row width / leading dimension");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long blocks = (n + WPB - 1) / WPB;
    const dim3 grid((unsigned)(blocks > 4096 ? 4096 : blocks)), block(256);
#define EGK_GR(S, D) hipLaunchKernelGGL((gather_rows_kernel<S, D>), grid, block, 0, s, (const S*)table, (long long)ld, (const long long*)idx, (D*)out, (long long)n, cols, (long long)table_rows)
    if (table_dtype == EGK_F32 && out_dtype == EGK_F32) EGK_GR(float, float);
    else if (table_dtype == EGK_F32 && out_dtype == EGK_BF16) EGK_GR(float, bf16_t);
    else if (table_dtype == EGK_BF16 && out_dtype == EGK_BF16) EGK_GR(bf16_t, bf16_t);
    else if (table_dtype == EGK_BF16 && out_dtype == EGK_F32) EGK_GR(bf16_t, float);
    else {
        set_error("egk_gather_rows: unknown element type");
        return EGK_EINVAL;
    }
#undef EGK_GR
    return check_launch("egk_gather_rows");
}

int egk_gather_lerp_rows(egk_stream_t stream, const void* table, int32_t table_dtype, int64_t ld, int64_t table_rows,
                         const int64_t* lo, const int64_t* hi, const double* w, void* out, int32_t out_dtype, int64_t n,
                         int32_t cols) {
    EGK_REQUIRE(table && lo && hi && w && out, "egk_gather_lerp_rows: null pointer");
    EGK_REQUIRE(cols >= 1 && ld >= cols, "egk_gather_lerp_rows: bad row width / leading dimension");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long blocks = (n + WPB - 1) / WPB;
    const dim3 grid((unsigned)(blocks > 4096 ? 4096 : blocks)), block(256);
#define EGK_GL(S, D) hipLaunchKernelGGL((gather_lerp_rows_kernel<S, D>), grid, block, 0, s, (const S*)table, (long long)ld, (const long long*)lo, (const long long*)hi, w, (D*)out, (long long)n, cols, (long long)table_rows)
    if (table_dtype == EGK_F32 && out_dtype == EGK_F32) EGK_GL(float, float);
    else if (table_dtype == EGK_F32 && out_dtype == EGK_BF16) EGK_GL(float, bf16_t);
    else if (table_dtype == EGK_BF16 && out_dtype == EGK_BF16) EGK_GL(bf16_t, bf16_t);
    else if (table_dtype == EGK_BF16 && out_dtype == EGK_F32) EGK_GL(bf16_t, float);
    else {
        set_error("egk_gather_lerp_rows: unknown element type");
        return EGK_EINVAL;
    }
#undef EGK_GL
    return check_launch("egk_gather_lerp_rows");
}

int egk_pe_add(egk_stream_t stream, const void* x, const int64_t* pos, const float* freq, void* y, int32_t rows,
               int32_t cols, int32_t dtype) {
    EGK_REQUIRE(x && pos && freq && y, "egk_pe_add: null pointer");
    EGK_REQUIRE((cols & 1) == 0, "egk_pe_add: odd channel count");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_PE_ADD, s, 0, (dtype == EGK_BF16 ? 4.0 : 8.0) * rows * cols);
    const SplitTee tee = take_split_tee();
    EGK_REQUIRE(!tee.lo || dtype == EGK_F32, "egk_pe_add: a split tee needs an f32 result");
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(pe_add_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, s, (const T*)x,
                                             (const long long*)pos, freq, (T*)y, rows, cols, tee));
    return check_launch("egk_pe_add");
}

int egk_pe_table(egk_stream_t stream, const float* freq, int64_t pos_min, int32_t n_pos, int32_t cols, float* table) {
    EGK_REQUIRE(freq && table && n_pos >= 1, "egk_pe_table: bad arguments");
    EGK_REQUIRE((cols & 1) == 0, "egk_pe_table: odd channel count");
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)n_pos * cols;
    hipLaunchKernelGGL(pe_table_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, s, freq,
                       (long long)pos_min, n_pos, cols, table);
    return check_launch("egk_pe_table");
}

int egk_pe_add_table(egk_stream_t stream, const void* x, const int64_t* pos, const float* freq, const float* table, int64_t pos_min,
                     int32_t n_pos, void* y, int32_t rows, int32_t cols, int32_t dtype) {
    EGK_REQUIRE(x && pos && freq && table && y && n_pos >= 1, "egk_pe_add_table: null pointer");
    EGK_REQUIRE((cols & 1) == 0, "egk_pe_add_table: odd channel count");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_PE_ADD, s, 0, (dtype == EGK_BF16 ? 4.0 : 8.0) * rows * cols);
    const SplitTee tee = take_split_tee();
    EGK_REQUIRE(!tee.lo || dtype == EGK_F32, "egk_pe_add_table: a split tee needs an f32 result");
    EGK_REQUIRE(!slab_input_armed() || dtype == EGK_F32, "egk_pe_add_table: a slab input needs an f32 input");
    if (g_graph_rows_v2 && dtype == EGK_BF16 && cols == 1024 && al16(x) && al16(y) && al16(table)) {
        const int g2 = cdiv(rows, 2 * WPB) > 1024 ? 1024 : cdiv(rows, 2 * WPB);
        hipLaunchKernelGGL(pe_add_table_1k_kernel<2>, dim3(g2), dim3(256), 0, s, (const bf16_t*)x, (const long long*)pos, freq, table,
                           (long long)pos_min, n_pos, (bf16_t*)y, rows);
        return check_launch("egk_pe_add_table");
    }
    const SlabInput si = take_slab_input();
    EGK_REQUIRE(!si.x2 || (dtype == EGK_F32 && cols % 4 == 0), "egk_pe_add_table: a slab input needs f32 rows of a multiple of 4 columns");
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(pe_add_table_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, s, (const T*)x,
                                             (const long long*)pos, freq, table, (long long)pos_min, n_pos, (T*)y, rows, cols, tee, si));
    return check_launch("egk_pe_add_table");
}

int64_t egk_csr_heavy_ws_bytes(int32_t n_heavy, int32_t cols) { return (int64_t)n_heavy * CSR_CHUNKS * cols * 4; }

int32_t egk_csr_heavy_threshold(void) { return VERY_HEAVY; }

static int csr_gather_impl(egk_stream_t stream, const void* x, const int32_t* rowptr, const int32_t* col, const float* wgt,
                           const void* relu_gate, void* out, int32_t rows, int32_t cols, int32_t dtype, const int32_t* heavy_rows,
                           int32_t n_heavy, float* ws, int32_t heavy_mode, const unsigned char* band) {
    EGK_REQUIRE(x && rowptr && out, "egk_csr_gather: null pointer");
    EGK_REQUIRE(!band || !wgt, "egk_csr_gather_banded: the neighbour codes describe the unweighted (mean) orientation only");
    EGK_REQUIRE(n_heavy == 0 || (heavy_rows && (ws || heavy_mode == 1)), "egk_csr_gather: heavy rows need their list and a workspace");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_CSR_GATHER, s, 0, (dtype == EGK_BF16 ? 0.5 : 1.0) * (relu_gate ? 12.0 : 8.0) * rows * cols);
    EGK_REQUIRE(cols <= 4096, "egk_csr_gather: rows wider than 4096 are unsupported");
    const int skip_above = n_heavy > 0 ? VERY_HEAVY : 0x7fffffff;
    const int in_launch = (n_heavy > 0 && heavy_mode == 1) ? n_heavy : 0;  // listed rows summed by one workgroup each, in the same launch
    const bool lean_band = g_graph_rows_v2 && band && !wgt && !relu_gate && !in_launch && !split_tee_armed() && dtype == EGK_BF16 &&
                           cols == 1024 && al16(x) && al16(out);
    if (lean_band) {  // (listed rows are skipped and produced by the split launches below)
        EGK_REQUIRE(col, "egk_csr_gather_banded: null column indices");
        const int g2 = cdiv(rows, 2 * WPB) > 1024 ? 1024 : cdiv(rows, 2 * WPB);
        hipLaunchKernelGGL(csr_gather_band_1k_kernel<2>, dim3(g2), dim3(256), 0, s, (const bf16_t*)x, rowptr, col, band, (bf16_t*)out, rows,
                           skip_above);
        if (n_heavy == 0) return check_launch("egk_csr_gather");
    }
    const bool lean = !lean_band && g_graph_rows_v2 && !band && !split_tee_armed() && dtype == EGK_BF16 && cols == 1024 && col && al16(x) &&
                      al16(out) && (!relu_gate || al16(relu_gate));
    if (lean) {  // (listed rows that are not summed in this launch are skipped by it and produced by the split launches below)
        const int g1 = (cdiv(rows, WPB) > 2048 ? 2048 : cdiv(rows, WPB)) + in_launch;
#define EGK_CSR1K(W, G) hipLaunchKernelGGL((csr_gather_1k_kernel<W, G>), dim3(g1), dim3(256), 0, s, (const bf16_t*)x, rowptr, col, wgt, (const bf16_t*)relu_gate, (bf16_t*)out, rows, skip_above, heavy_rows, in_launch)
        if (wgt) { if (relu_gate) EGK_CSR1K(true, true); else EGK_CSR1K(true, false); }
        else { if (relu_gate) EGK_CSR1K(false, true); else EGK_CSR1K(false, false); }
#undef EGK_CSR1K
        if (n_heavy == 0 || in_launch) return check_launch("egk_csr_gather");
    }
    const SplitTee tee = (lean || lean_band) ? SplitTee{nullptr, nullptr, 0} : take_split_tee();
    EGK_REQUIRE(!tee.lo || dtype == EGK_F32, "egk_csr_gather: a split tee needs an f32 result");
    EGK_REQUIRE(!tee.lo || n_heavy == 0 || in_launch, "egk_csr_gather: no split tee with rows finished by the split launches");
#define EGK_CSR(NVV) hipLaunchKernelGGL((csr_gather_kernel<NVV, T>), dim3(row_grid(rows) + in_launch), dim3(256), 3 * NVV * 256 * sizeof(float), s, (const T*)x, rowptr, col, wgt, (const T*)relu_gate, (T*)out, rows, cols, skip_above, heavy_rows, in_launch, band, tee)
    if (!lean && !lean_band) EGK_DISPATCH_T(dtype, { if (cols <= 256) EGK_CSR(1); else if (cols <= 1024) EGK_CSR(4); else EGK_CSR(16); });
#undef EGK_CSR
    if (n_heavy > 0 && !in_launch) {
        EGK_DISPATCH_T(dtype, {
            hipLaunchKernelGGL((csr_heavy_partial_kernel<T>), dim3(n_heavy, CSR_CHUNKS), dim3(256), 0, s, (const T*)x, rowptr, col,
                               wgt, heavy_rows, ws, cols);
            hipLaunchKernelGGL((csr_heavy_finish_kernel<T>), dim3(n_heavy), dim3(256), 0, s, (const float*)ws, rowptr, wgt,
                               (const T*)relu_gate, heavy_rows, (T*)out, cols);
        });
    }
    return check_launch("egk_csr_gather");
}

int egk_csr_gather(egk_stream_t stream, const void* x, const int32_t* rowptr, const int32_t* col, const float* wgt,
                   const void* relu_gate, void* out, int32_t rows, int32_t cols, int32_t dtype, const int32_t* heavy_rows,
                   int32_t n_heavy, float* ws, int32_t heavy_mode) {
    return csr_gather_impl(stream, x, rowptr, col, wgt, relu_gate, out, rows, cols, dtype, heavy_rows, n_heavy, ws, heavy_mode, nullptr);
}

int egk_csr_gather_banded(egk_stream_t stream, const void* x, const int32_t* rowptr, const int32_t* col, const uint8_t* band,
                          void* out, int32_t rows, int32_t cols, int32_t dtype, const int32_t* heavy_rows, int32_t n_heavy,
                          float* ws, int32_t heavy_mode) {
    EGK_REQUIRE(band, "egk_csr_gather_banded: null neighbour codes");
    return csr_gather_impl(stream, x, rowptr, col, nullptr, nullptr, out, rows, cols, dtype, heavy_rows, n_heavy, ws, heavy_mode, band);
}

int egk_gather_max_tune(int32_t up_front) {
    const int prev = g_gather_max_up_front;
    if (up_front >= 0) g_gather_max_up_front = up_front != 0;
    return prev;
}

int egk_gather_max_group_fwd(egk_stream_t stream, const void* f, const float* const* banks, const int64_t* const* nns,
                             int32_t n_groups, void* m, uint8_t* arg, int32_t rows, int32_t cols, int32_t k, int32_t dtype) {
    EGK_REQUIRE(f && banks && nns && m && arg, "egk_gather_max_group_fwd: null pointer");
    EGK_REQUIRE(n_groups >= 1 && n_groups <= 4, "egk_gather_max_group_fwd: 1 .. 4 groups");
    EGK_REQUIRE(k >= 0 && k < 255, "egk_gather_max_group_fwd: k out of range");
    for (int g = 0; g < n_groups; ++g) EGK_REQUIRE(banks[g] && nns[g], "egk_gather_max_group_fwd: null pointer");
    if (rows == 0 || cols == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    GatherMaxGroups gg;
    gg.n_groups = n_groups; gg.rows = rows;
    for (int g = 0; g < 4; ++g) {
        gg.bank[g] = banks[g < n_groups ? g : 0];
        gg.nn[g] = (const long long*)nns[g < n_groups ? g : 0];
    }
    const int ebytes = dtype == EGK_BF16 ? 2 : 4;
    {
        ProfScope prof(KID_GATHER_MAX_FWD, s, 0, (4.0 * (k + 2) + 1.0) * rows * cols * n_groups);
        bool done = false;
        EGK_DISPATCH_T(dtype, done = gather_max_up_front<T>(s, (const T*)f, gg, (T*)m, arg, cols, k));
        if (done) return check_launch("egk_gather_max_group_fwd");
    }
    for (int g = 0; g < n_groups; ++g) {  // shapes the grouped kernel does not take: one generic launch per group
        const long long off = (long long)g * rows * cols;
        const int rc = egk_gather_max_fwd(stream, (const char*)f + off * ebytes, banks[g], nns[g], (char*)m + off * ebytes, arg + off,
                                          rows, cols, k, dtype);
        if (rc) return rc;
    }
    return 0;
}

int egk_gather_max_fwd(egk_stream_t stream, const void* f, const float* bank, const int64_t* nn, void* m, uint8_t* arg,
                       int32_t rows, int32_t cols, int32_t k, int32_t dtype) {
    EGK_REQUIRE(f && bank && nn && m && arg, "egk_gather_max_fwd: null pointer");
    EGK_REQUIRE(k >= 0 && k < 255, "egk_gather_max_fwd: k out of range");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_GATHER_MAX_FWD, s, 0, (4.0 * (k + 2) + 1.0) * rows * cols);
    {
        GatherMaxGroups gg;
        gg.n_groups = 1; gg.rows = rows;
        for (int g = 0; g < 4; ++g) { gg.bank[g] = bank; gg.nn[g] = (const long long*)nn; }
        bool done = false;
        EGK_DISPATCH_T(dtype, done = gather_max_up_front<T>(s, (const T*)f, gg, (T*)m, arg, cols, k));
        if (done) return check_launch("egk_gather_max_fwd");
    }
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(gather_max_fwd_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, s, (const T*)f, bank,
                                             (const long long*)nn, (T*)m, arg, rows, cols, k));
    return check_launch("egk_gather_max_fwd");
}

int egk_gather_max_bwd(egk_stream_t stream, const void* dm, const uint8_t* arg, void* df, int32_t rows, int32_t cols,
                       int32_t k, int32_t accumulate, int32_t dtype) {
    EGK_REQUIRE(dm && arg && df, "egk_gather_max_bwd: null pointer");
    const long long n = (long long)rows * cols;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_GATHER_MAX_BWD, s, 0, 9.0 * n);
    const long long blocks = (n / 4 + 255) / 256;
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(gather_max_bwd_kernel<T>, dim3((unsigned)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks)),
                                             dim3(256), 0, s, (const T*)dm, arg, (T*)df, n, k, accumulate));
    return check_launch("egk_gather_max_bwd");
}

int egk_segment_max_fwd(egk_stream_t stream, const void* x, const int32_t* ptr, void* out, int32_t* arg, int32_t n_seg,
                        int32_t cols, int32_t dtype) {
    EGK_REQUIRE(x && ptr && out && arg, "egk_segment_max_fwd: null pointer");
    const void* xs[1] = {x};
    void* outs[1] = {out};
    int32_t* args[1] = {arg};
    return egk_segment_max_multi_fwd(stream, xs, ptr, outs, args, 1, n_seg, cols, dtype);
}

int egk_segment_max_multi_fwd(egk_stream_t stream, const void* const* xs, const int32_t* ptr, void* const* outs, int32_t* const* args,
                              int32_t n_src, int32_t n_seg, int32_t cols, int32_t dtype) {
    EGK_REQUIRE(xs && ptr && outs && args, "egk_segment_max_multi_fwd: null pointer");
    EGK_REQUIRE(n_src >= 1 && n_src <= 4, "egk_segment_max_multi_fwd: 1..4 inputs");
    if (n_seg == 0 || cols == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_SEGMAX_FWD, s, 0, 0);
    const int ebytes = dtype == EGK_BF16 ? 2 : 4;
    bool v4 = cols % 4 == 0;
    for (int i = 0; i < n_src; ++i) {
        EGK_REQUIRE(xs[i] && outs[i] && args[i], "egk_segment_max_multi_fwd: null input pointer");
        v4 = v4 && (reinterpret_cast<uintptr_t>(xs[i]) % (4 * ebytes)) == 0 && (reinterpret_cast<uintptr_t>(outs[i]) % (4 * ebytes)) == 0 &&
             (reinterpret_cast<uintptr_t>(args[i]) & 15) == 0;
    }
    if (v4) {
        EGK_DISPATCH_T(dtype, {
            SegMaxSrcs<T> src;
            for (int i = 0; i < 4; ++i) {
                const int j = i < n_src ? i : 0;
                src.x[i] = (const T*)xs[j];
                src.out[i] = (T*)outs[j];
                src.arg[i] = args[j];
            }
            if (n_seg < 128)  // (few segments: sixteen row lanes per workgroup)
                hipLaunchKernelGGL((segmax_fwd_v4_kernel<T, 16>), dim3(cdiv(cols, 256), n_seg, n_src), dim3(1024), 0, s, src, ptr, n_seg, cols);
            else
                hipLaunchKernelGGL((segmax_fwd_v4_kernel<T, 4>), dim3(cdiv(cols, 256), n_seg, n_src), dim3(256), 0, s, src, ptr, n_seg, cols);
        });
        return check_launch("egk_segment_max_fwd");
    }
    for (int i = 0; i < n_src; ++i)
        EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(segmax_fwd_kernel<T>, dim3(cdiv(cols, 256), n_seg), dim3(256), 0, s, (const T*)xs[i], ptr,
                                                 (T*)outs[i], args[i], n_seg, cols));
    return check_launch("egk_segment_max_fwd");
}

int egk_segment_max_bwd(egk_stream_t stream, const void* dout, const int32_t* arg, const int32_t* ptr, void* dx,
                        int32_t n_seg, int32_t rows, int32_t cols, int32_t dtype) {
    EGK_REQUIRE(dout && arg && ptr && dx, "egk_segment_max_bwd: null pointer");
    const void* douts[1] = {dout};
    const int32_t* args[1] = {arg};
    void* dxs[1] = {dx};
    return egk_segment_max_multi_bwd(stream, douts, args, ptr, dxs, 1, n_seg, rows, cols, dtype);
}

int egk_segment_max_multi_bwd(egk_stream_t stream, const void* const* douts, const int32_t* const* args, const int32_t* ptr,
                              void* const* dxs, int32_t n_src, int32_t n_seg, int32_t rows, int32_t cols, int32_t dtype) {
    EGK_REQUIRE(douts && args && ptr && dxs, "egk_segment_max_multi_bwd: null pointer");
    EGK_REQUIRE(n_src >= 1 && n_src <= 4, "egk_segment_max_multi_bwd: 1..4 inputs");
    if (n_seg == 0 || cols == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_SEGMAX_BWD, s, 0, 0);
    for (int i = 0; i < n_src; ++i) EGK_REQUIRE(douts[i] && args[i] && dxs[i], "egk_segment_max_multi_bwd: null input pointer");
    EGK_DISPATCH_T(dtype, {
        SegMaxGrads<T> gr;
        for (int i = 0; i < 4; ++i) {
            const int j = i < n_src ? i : 0;
            gr.dout[i] = (const T*)douts[j];
            gr.arg[i] = args[j];
            gr.dx[i] = (T*)dxs[j];
        }
        hipLaunchKernelGGL(segmax_bwd_kernel<T>, dim3(cdiv(cols, 256), n_seg, n_src), dim3(256), 0, s, gr, ptr, n_seg, cols);
    });
    return check_launch("egk_segment_max_bwd");
}

int egk_row_inv_norm(egk_stream_t stream, const void* x, float* inv_norm, int32_t rows, int32_t cols, int32_t dtype) {
    EGK_REQUIRE(x && inv_norm, "egk_row_inv_norm: null pointer");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_ROW_INV_NORM, s, 0, 4.0 * rows * cols);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(row_inv_norm_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, s, (const T*)x, inv_norm,
                                             rows, cols, 0));
    return check_launch("egk_row_inv_norm");
}

int egk_row_inv_norm_cast(egk_stream_t stream, const float* x, float* inv_norm, void* hi, void* h16, int32_t rows, int32_t cols) {
    EGK_REQUIRE(x && inv_norm && hi, "egk_row_inv_norm_cast: null pointer");
    EGK_REQUIRE(cols % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)hi % 8 == 0) && ((uintptr_t)h16 % 8 == 0),
                "egk_row_inv_norm_cast: rows of a multiple of 4 columns, 16-byte aligned input, 8-byte aligned outputs");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_ROW_INV_NORM, s, 0, (h16 ? 8.0 : 6.0) * rows * cols);
    hipLaunchKernelGGL(row_inv_norm_cast_kernel, dim3(row_grid(rows)), dim3(256), 0, s, x, inv_norm, (bf16_t*)hi, (unsigned short*)h16,
                       rows, cols);
    return check_launch("egk_row_inv_norm_cast");
}

int egk_row_sq_norm(egk_stream_t stream, const void* x, float* sq_norm, int32_t rows, int32_t cols, int32_t dtype) {
    EGK_REQUIRE(x && sq_norm, "egk_row_sq_norm: null pointer");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_ROW_INV_NORM, s, 0, 4.0 * rows * cols);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(row_inv_norm_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, s, (const T*)x, sq_norm,
                                             rows, cols, 1));
    return check_launch("egk_row_sq_norm");
}

int egk_cos_dist(egk_stream_t stream, const float* dot, int64_t ldd, const float* f_inv, const float* b_inv, float* dist,
                 int32_t rows, int32_t K) {
    EGK_REQUIRE(dot && f_inv && b_inv && dist, "egk_cos_dist: null pointer");
    if (rows == 0 || K == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_TOPK, s, 0, 8.0 * rows * K);
    hipLaunchKernelGGL(cos_dist_kernel, dim3(cdiv(K, 256), rows), dim3(256), 0, s, dot, (long long)ldd, f_inv, b_inv, dist,
                       rows, K);
    return check_launch("egk_cos_dist");
}

static int topk_launch(const char* what, bool l2, egk_stream_t stream, const float* dot, int64_t ldd, const float* fa,
                       const float* ba, int64_t* nn, int32_t rows, int32_t K, int32_t k) {
    EGK_REQUIRE(dot && fa && ba && nn, "%s: null pointer", what);
    EGK_REQUIRE(k >= 1 && k <= 16 && k <= K, "%s: k must be in [1, min(16, K)]", what);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_TOPK, s, 0, 4.0 * rows * K);
    const int vec = (ldd % 4 == 0) && ((uintptr_t)dot % 16 == 0) && ((uintptr_t)ba % 16 == 0);
    const dim3 grid(row_grid(rows)), block(256);
#define EGK_TOPK(KM, L2) \
    hipLaunchKernelGGL((topk_kernel<KM, L2>), grid, block, 0, s, dot, (long long)ldd, fa, ba, (long long*)nn, rows, K, k, vec)
    if (k <= 4) {
        if (l2) EGK_TOPK(4, true); else EGK_TOPK(4, false);
    } else {
        if (l2) EGK_TOPK(16, true); else EGK_TOPK(16, false);
    }
#undef EGK_TOPK
    return check_launch(what);
}

int egk_topk_smallest(egk_stream_t stream, const float* dot, int64_t ldd, const float* f_inv, const float* b_inv,
                      int64_t* nn, int32_t rows, int32_t K, int32_t k) {
    return topk_launch("egk_topk_smallest", false, stream, dot, ldd, f_inv, b_inv, nn, rows, K, k);
}

int egk_topk_smallest_l2(egk_stream_t stream, const float* dot, int64_t ldd, const float* f_sq, const float* b_sq,
                         int64_t* nn, int32_t rows, int32_t K, int32_t k) {
    return topk_launch("egk_topk_smallest_l2", true, stream, dot, ldd, f_sq, b_sq, nn, rows, K, k);
}

int egk_bf16_residual_ratio(egk_stream_t stream, const float* x, int64_t ld, float* r, float* rmax, int32_t rows, int32_t cols) {
    return egk_residual_ratio16(stream, x, ld, r, rmax, rows, cols, 0);
}

int egk_residual_ratio16(egk_stream_t stream, const float* x, int64_t ld, float* r, float* rmax, int32_t rows, int32_t cols, int32_t f16) {
    EGK_REQUIRE(x && r && rmax, "egk_residual_ratio16: null pointer");
    EGK_REQUIRE(rows >= 1 && cols >= 1 && ld >= cols, "egk_residual_ratio16: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (f16) hipLaunchKernelGGL(bf16_residual_ratio_kernel<true>, dim3(row_grid(rows)), dim3(256), 0, s, x, (long long)ld, r, rows, cols);
    else hipLaunchKernelGGL(bf16_residual_ratio_kernel<false>, dim3(row_grid(rows)), dim3(256), 0, s, x, (long long)ld, r, rows, cols);
    hipLaunchKernelGGL(max_of_kernel, dim3(1), dim3(256), 0, s, r, rmax, rows);
    return check_launch("egk_residual_ratio16");
}

int egk_cast_f16(egk_stream_t stream, const float* x, void* y, int64_t n) {
    EGK_REQUIRE(x && y && n >= 0, "egk_cast_f16: bad arguments");
    EGK_REQUIRE(((reinterpret_cast<uintptr_t>(x) & 15) | (reinterpret_cast<uintptr_t>(y) & 7)) == 0, "egk_cast_f16: unaligned buffers");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(cast_f16_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, x, (unsigned short*)y,
                       (long long)n);
    return check_launch("egk_cast_f16");
}

int egk_topk_window(egk_stream_t stream, const float* dot1, int64_t ldd, const float* f, int64_t ldf, const float* bank, int64_t ldb,
                    const float* f_inv, const float* b_inv, const float* rb_max, int64_t* nn, int32_t* cand, int32_t rows, int32_t K,
                    int32_t H, int32_t k) {
    EGK_REQUIRE(dot1 && f && bank && f_inv && b_inv && rb_max && nn, "egk_topk_window: null pointer");
    EGK_REQUIRE(k >= 1 && k <= 16 && k <= K, "egk_topk_window: k must be in [1, min(16, K)]");
    EGK_REQUIRE(H >= 1 && ldf >= H && ldb >= H && ldd >= K, "egk_topk_window: bad leading dimension");
    if (rows == 0) return 0;
    const float* banks[1] = {bank};
    const float* b_invs[1] = {b_inv};
    const float* rb_maxs[1] = {rb_max};
    return egk_topk_window_group(stream, dot1, ldd, f, ldf, banks, ldb, f_inv, b_invs, rb_maxs, nn, cand, 1, rows, K, H, k);
}

int egk_topk_window_group(egk_stream_t stream, const float* dot1, int64_t ldd, const float* f, int64_t ldf, const float* const* banks,
                          int64_t ldb, const float* f_inv, const float* const* b_invs, const float* const* rb_maxs, int64_t* nn,
                          int32_t* cand, int32_t n_groups, int32_t rows_per_group, int32_t K, int32_t H, int32_t k) {
    return egk_topk_window_group16(stream, dot1, ldd, f, ldf, banks, ldb, f_inv, b_invs, rb_maxs, nn, cand, n_groups, rows_per_group, K, H, k, 0);
}

int egk_topk_window_group16(egk_stream_t stream, const float* dot1, int64_t ldd, const float* f, int64_t ldf, const float* const* banks,
                            int64_t ldb, const float* f_inv, const float* const* b_invs, const float* const* rb_maxs, int64_t* nn,
                            int32_t* cand, int32_t n_groups, int32_t rows_per_group, int32_t K, int32_t H, int32_t k, int32_t screen_f16) {
    EGK_REQUIRE(dot1 && f && banks && f_inv && b_invs && rb_maxs && nn, "egk_topk_window_group: null pointer");
    EGK_REQUIRE(n_groups >= 1 && n_groups <= 8, "egk_topk_window_group: 1..8 groups");
    EGK_REQUIRE(k >= 1 && k <= 16 && k <= K, "egk_topk_window_group: k must be in [1, min(16, K)]");
    EGK_REQUIRE(H >= 1 && ldf >= H && ldb >= H && ldd >= K, "egk_topk_window_group: bad leading dimension");
    if (rows_per_group == 0) return 0;
    EGK_REQUIRE(rows_per_group > 0, "egk_topk_window_group: negative row count");
    const int rows = n_groups * rows_per_group;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_TOPK, s, 0, 8.0 * rows * K);
    TopkWindowBanks tb;
    int vec = (ldd % 4 == 0) && ((uintptr_t)dot1 % 16 == 0);
    for (int g = 0; g < 8; ++g) {
        const int src = g < n_groups ? g : 0;
        EGK_REQUIRE(banks[src] && b_invs[src] && rb_maxs[src], "egk_topk_window_group: null bank pointer");
        EGK_REQUIRE((uintptr_t)banks[src] % 16 == 0, "egk_topk_window_group: bank rows must be 16-byte aligned");
        tb.bank[g] = banks[src];
        tb.b_inv[g] = b_invs[src];
        tb.rb_max[g] = rb_maxs[src];
        vec = vec && ((uintptr_t)b_invs[src] % 16 == 0);
    }
    tb.rows_per_group = rows_per_group;
    tb.screen_f16 = screen_f16 ? 1 : 0;
    EGK_REQUIRE((uintptr_t)f % 16 == 0, "egk_topk_window_group: feature rows must be 16-byte aligned");
    const dim3 grid(row_grid(rows)), block(256);
    if (k <= 4)
        hipLaunchKernelGGL((topk_window_kernel<4>), grid, block, 0, s, dot1, (long long)ldd, f, (long long)ldf, tb, (long long)ldb, f_inv,
                           (long long*)nn, cand, rows, K, H, k, vec);
    else
        hipLaunchKernelGGL((topk_window_kernel<16>), grid, block, 0, s, dot1, (long long)ldd, f, (long long)ldf, tb, (long long)ldb, f_inv,
                           (long long*)nn, cand, rows, K, H, k, vec);
    return check_launch("egk_topk_window_group");
}

int egk_segment_sum_rows_f64(egk_stream_t stream, const void* x, const int32_t* order, const int32_t* seg_ptr,
                             const int64_t* seg_label, double* bank, int64_t* count, int32_t n_seg, int32_t cols,
                             int64_t n_labels, int32_t dtype) {
    EGK_REQUIRE(x && order && seg_ptr && seg_label && bank, "egk_segment_sum_rows_f64: null pointer");
    if (n_seg == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_SCATTER_ADD_F64, s, 0, 4.0 * n_seg * cols + 16.0 * n_seg * cols);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(segment_sum_f64_kernel<T>, dim3(row_grid(n_seg)), dim3(256), 0, s, (const T*)x, order,
                                             seg_ptr, (const long long*)seg_label, bank, (long long*)count, n_seg, cols,
                                             (long long)n_labels));
    return check_launch("egk_segment_sum_rows_f64");
}

int egk_gather_max_bank_grad(egk_stream_t stream, const void* dm, const uint8_t* arg, const int32_t* t_rowptr,
                             const int32_t* t_edge, float* dbank, int32_t K, int32_t cols, int32_t k, int32_t dtype) {
    EGK_REQUIRE(dm && arg && t_rowptr && t_edge && dbank, "egk_gather_max_bank_grad: null pointer");
    EGK_REQUIRE(k >= 1 && k < 255, "egk_gather_max_bank_grad: k out of range");
    if (K == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_GATHER_MAX_BWD, s, 0, 8.0 * K * cols);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(gather_max_bank_grad_kernel<T>, dim3(row_grid(K)), dim3(256), 0, s, (const T*)dm, arg,
                                             t_rowptr, t_edge, dbank, K, cols, k));
    return check_launch("egk_gather_max_bank_grad");
}
}
