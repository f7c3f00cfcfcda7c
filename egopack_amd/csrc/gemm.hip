// MFMA contraction kernel for gfx950: C = epilogue(op(A) . op(B)^T), two-source K, split-K slabs.
//
// Geometry (one workgroup = 256 threads = 4 waves as 2(M) x 2(N); one output tile 128 x 128):
//   * both operands are staged as K-contiguous LDS images  [128 rows][128 B]  whatever their
//     layout and element type in memory:
//       - row-major f32   : 2 x 16-B loads per lane along K, converted to bf16 while staging
//       - row-major bf16  : 1 x 16-B load per lane along K (no conversion: the fast path)
//       - transposed f32  ([K][rows]): 16 B per lane along the contiguous "rows" axis, 4x8 register transpose
//       - transposed bf16 ([K][rows]): 8 x 16-B loads per lane, 8x8 16-bit register transpose (v_perm_b32)
//     so no transposed copies of weights, activations or gradients ever exist in HBM;
//   * EGK_COMPUTE_BF16: v_mfma_f32_16x16x32_bf16 (64-element K-tile); EGK_COMPUTE_F32:
//     v_mfma_f32_16x16x4_f32 on f32 operands (exact fmaf chain, 32-element K-tile).  One 16-byte LDS chunk
//     per lane feeds one bf16 MFMA or four f32 MFMAs;
//   * the 16-byte chunk index of a row is XOR-swizzled with (row>>1)&7 so that the ds_read_b128 of a
//     16-lane group (16 rows, one chunk column) hits 16 distinct 16-B slots of the 256-B bank row;
//   * register prefetch: the global loads of K-tile t+1 are issued before the MFMAs of tile t and written
//     to LDS after the barrier (T14 split), single LDS image of 32 KiB;
//   * the MFMA is issued as D^T = B.A^T so each lane owns 4 CONSECUTIVE columns of one output row: bias /
//     residual / C traffic is 16 B (f32) or 8 B (bf16) per lane;
//   * blockIdx is remapped so that each XCD owns a contiguous range of M-tiles.
#include <type_traits>

#include "common.h"

namespace egk {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned short bf16_t;  // storage type of a bf16 element in memory

constexpr int BM = 128, BN = 128, ROWB = 128;  // LDS row = 128 bytes = 8 chunks of 16 B
constexpr int NTHREADS = 256;

constexpr int MAXSRC = 6;  // K sources of one contraction: C = sum_s op(A_s) . op(B_s)^T (same layout / element type)
struct GemmArgs {
    int M, N;
    int nsrc;
    int K[MAXSRC];
    const void* A[MAXSRC];
    const void* B[MAXSRC];
    long long lda[MAXSRC], ldb[MAXSRC];
    int a_vec, b_vec;  // bit s: 16-byte vector loads legal for source s
    void* C;
    long long ldc;
    int c_vec, c_bf16;
    int accumulate, act;
    float alpha;
    const float* bias;
    const void* residual;
    long long ldr;
    int r_vec, r_bf16;
    int splitk;
    float* ws;  // [splitk][M][N] partial slabs when splitk > 1
    int tiles_m, tiles_n;
    int group_m;      // pipelined kernel: tile rows per group of the XCD-local tile order (see tile_of)
    float* dbias;     // fused bias gradient: dbias[m] += sum_k op(A)[m, k]  (transA pipelined kernel only)
    float* ws_bias;   // [splitk][M] partial row sums when splitk > 1
    int rows_epilogue;  // 4-wave pipelined variants: write the tile out through LDS in whole rows (development knob, default 1)
    // per-segment sums of the RESULT for the graph-mode LayerNorm that consumes it (egk_gemm_desc.st_*), rows epilogue only
    int st_mode, st_nseg;
    const int* st_seg_ptr;
    double* st_ws;             // [tiles_m * tiles_n][st_nseg][2]
    const void* st_x;          // mode 2: the LayerNorm's input (same shape / element type as C), leading dimension st_ldx
    long long st_ldx;
    const float* st_stats;     // mode 2: [st_nseg][2] (mean, 1 / (std + eps)) of the forward pass
    const float* st_w;
    const float* st_b;
    float st_slope;
};

// Workgroup -> (slab z, tile row, tile column) for a 1-D launch of tiles_m * tiles_n * splitk workgroups.
// The hardware deals consecutive workgroup ids round-robin over the 8 XCDs; each XCD has its own L2.  XCD x is
// given a CONSECUTIVE run of the (z-major) virtual ids, and virtual ids walk the tile grid in groups of group_m tile
// rows, column by column inside a group: the run of one XCD is a near-square patch of ONE slab, so the operand
// strips its workgroups fetch (one A strip per tile row, one B strip per tile column) overlap as much as they can
// in that XCD's L2.  (Row-by-row order gives an XCD 1 x 36 tiles of the 8 x 36 dW grid: 37 strips instead of 12.)
// (virtual id -> tile: the slab-major walk in groups of group_m tile rows)
__device__ __forceinline__ void tile_of_virtual(const GemmArgs& g, int v, int& z, int& tm, int& tn) {
    const int tiles = g.tiles_m * g.tiles_n;
    z = v / tiles;
    const int t = v - z * tiles;
    const int gsz = g.group_m * g.tiles_n;
    const int grp = t / gsz, first = grp * g.group_m;
    const int rows = min(g.group_m, g.tiles_m - first);
    const int rr = t - grp * gsz;
    tm = first + rr % rows;
    tn = rr / rows;
}
__device__ __forceinline__ void tile_of(const GemmArgs& g, int bid, int& z, int& tm, int& tn) {
    const int tiles = g.tiles_m * g.tiles_n, nwg = tiles * g.splitk;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    z = v / tiles;
    const int t = v - z * tiles;
    const int gsz = g.group_m * g.tiles_n;
    const int grp = t / gsz, first = grp * g.group_m;
    const int rows = min(g.group_m, g.tiles_m - first);
    const int rr = t - grp * gsz;
    tm = first + rr % rows;
    tn = rr / rows;
}

// K tile t of the concatenated walk over the sources (tiles of KT elements) -> (source, tile inside it); uniform scalar work
template <int KT>
__device__ __forceinline__ void source_of(const GemmArgs& g, int t, int& src, int& tt) {
    src = 0;
    tt = t;
    while (src + 1 < g.nsrc) {
        const int n = (g.K[src] + KT - 1) / KT;
        if (tt < n) break;
        tt -= n;
        ++src;
    }
}
__device__ __forceinline__ int total_tiles(const GemmArgs& g, int KT) {
    int n = 0;
    for (int s = 0; s < g.nsrc; ++s) n += (g.K[s] + KT - 1) / KT;
    return n;
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint4 pack8(const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
    return __builtin_bit_cast(uint4, v);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// ---- staging loaders: produce this thread's share of a [128 rows][KT] K-contiguous tile -------------------------

// Row-major operand ([rows][K], ld): 4 slots: ids tid + 256*i -> (row = id>>3, chunk = id&7).
template <bool BF16C, typename T>
__device__ __forceinline__ void load_rowmajor(const T* __restrict__ base, long long ld, int rows_total, int row0, int k0,
                                              int klim, bool vec, uint4 (&out)[8]) {
    constexpr int CE = BF16C ? 8 : 4;  // elements per 16-byte LDS chunk
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + NTHREADS * i;
        const int r = id >> 3, c = id & 7;
        const int grow = row0 + r, gk = k0 + c * CE;
        const bool row_ok = grow < rows_total;
        const T* p = base + (long long)grow * ld + gk;
        if constexpr (sizeof(T) == 2) {  // bf16 in memory (BF16C only)
            if (row_ok && vec && gk + 8 <= klim) out[i] = *reinterpret_cast<const uint4*>(p);
            else {
                unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (row_ok && gk + j < klim) w[j >> 1] |= (unsigned)p[j] << (16 * (j & 1));
                out[i] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else {
            float f[8];
            if (row_ok && vec && gk + CE <= klim) {
                const float4 v0 = *reinterpret_cast<const float4*>(p);
                f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w;
                if constexpr (BF16C) {
                    const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
                    f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < CE; ++j) f[j] = (row_ok && gk + j < klim) ? p[j] : 0.0f;
            }
            if constexpr (BF16C) out[i] = pack8(f);
            else out[i] = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
        }
    }
}

// Transposed f32 operand ([K][rows], ld): thread -> row quad q = tid&31 (rows 4q..4q+3), chunk kc = tid>>5: 4 slots.
template <bool BF16C>
__device__ __forceinline__ void load_transposed_f32(const float* __restrict__ base, long long ld, int rows_total, int row0,
                                                    int k0, int klim, bool vec, uint4 (&out)[8]) {
    constexpr int CE = BF16C ? 8 : 4;
    const int tid = threadIdx.x;
    const int q = tid & 31, kc = tid >> 5;
    const int grow = row0 + 4 * q, gk = k0 + kc * CE;
    float f[CE][4];
#pragma unroll
    for (int j = 0; j < CE; ++j) {
        const float* p = base + (long long)(gk + j) * ld + grow;
        const bool k_ok = gk + j < klim;
        if (k_ok && vec && grow + 4 <= rows_total) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            f[j][0] = v.x; f[j][1] = v.y; f[j][2] = v.z; f[j][3] = v.w;
        } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) f[j][rr] = (k_ok && grow + rr < rows_total) ? p[rr] : 0.0f;
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        if constexpr (BF16C) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = f[j][rr];
            out[rr] = pack8(t);
        } else {
            out[rr] = make_uint4(__float_as_uint(f[0][rr]), __float_as_uint(f[1][rr]), __float_as_uint(f[2][rr]),
                                 __float_as_uint(f[3][rr]));
        }
    }
}

// Transposed bf16 operand ([K][rows], ld): an 8(k) x 8(rows) block per thread: row octet o = t&15, chunk kc = t>>4,
// t = tid & 127 -- only HALF of the workgroup (``half`` = 0: waves 0-1, 1: waves 2-3) stages it: 8 slots.
__device__ __forceinline__ void load_transposed_bf16(const bf16_t* __restrict__ base, long long ld, int rows_total, int row0,
                                                     int k0, int klim, bool vec, int half, uint4 (&out)[8]) {
    const int tid = threadIdx.x;
    if ((tid >> 7) != half) return;
    const int t = tid & 127;
    const int o = t & 15, kc = t >> 4;
    const int grow = row0 + 8 * o, gk = k0 + kc * 8;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bf16_t* p = base + (long long)(gk + j) * ld + grow;
        const bool k_ok = gk + j < klim;
        if (k_ok && vec && grow + 8 <= rows_total) v[j] = *reinterpret_cast<const uint4*>(p);
        else {
            unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
                if (k_ok && grow + rr < rows_total) w[rr >> 1] |= (unsigned)p[rr] << (16 * (rr & 1));
            v[j] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    // out[r].dword[e] = { lo: v[2e] elem r, hi: v[2e+1] elem r }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        unsigned d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint4& lo = v[2 * e];
            const uint4& hi = v[2 * e + 1];
            const unsigned s1 = (r >> 1) == 0 ? lo.x : (r >> 1) == 1 ? lo.y : (r >> 1) == 2 ? lo.z : lo.w;
            const unsigned s0 = (r >> 1) == 0 ? hi.x : (r >> 1) == 1 ? hi.y : (r >> 1) == 2 ? hi.z : hi.w;
            d[e] = __builtin_amdgcn_perm(s0, s1, (r & 1) ? 0x07060302u : 0x05040100u);
        }
        out[r] = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

// LDS store of what the matching loader produced.
template <bool TR, bool MEM16>
__device__ __forceinline__ void store_lds(unsigned char* lds, const uint4 (&v)[8], int half) {
    const int tid = threadIdx.x;
    if constexpr (TR && MEM16) {
        if ((tid >> 7) != half) return;
        const int t = tid & 127;
        const int o = t & 15, kc = t >> 4;
#pragma unroll
        for (int r = 0; r < 8; ++r) *reinterpret_cast<uint4*>(lds + lds_off(8 * o + r, kc)) = v[r];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row, chunk;
            if constexpr (TR) {
                row = 4 * (tid & 31) + i;
                chunk = tid >> 5;
            } else {
                const int id = tid + NTHREADS * i;
                row = id >> 3;
                chunk = id & 7;
            }
            *reinterpret_cast<uint4*>(lds + lds_off(row, chunk)) = v[i];
        }
    }
}

template <bool BF16C, bool TR, typename T>
__device__ __forceinline__ void load_operand(const void* base, long long ld, int rows_total, int row0, int k0, int klim,
                                             bool vec, int half, uint4 (&out)[8]) {
    if constexpr (TR && sizeof(T) == 2) load_transposed_bf16((const bf16_t*)base, ld, rows_total, row0, k0, klim, vec, half, out);
    else if constexpr (TR) load_transposed_f32<BF16C>((const float*)base, ld, rows_total, row0, k0, klim, vec, out);
    else load_rowmajor<BF16C, T>((const T*)base, ld, rows_total, row0, k0, klim, vec, out);
}

// Epilogue shared by the contraction kernels.  D^T layout: lane holds C[m = .. + lr][n = .. + 4*lg + t], t = 0..3.
template <int NI, int NJ>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x4 (&acc)[NI][NJ], int m0, int n0, int wm_off,
                                              int wn_off, int lr, int lg, int z) {
    const bool slab = g.splitk > 1;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = m0 + wm_off + i * 16 + lr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn_off + j * 16 + 4 * lg;
            if (n >= g.N) continue;
            f32x4 v = acc[i][j];
            const bool full = n + 4 <= g.N;
            if (slab) {
                float* p = g.ws + ((long long)z * g.M + m) * g.N + n;
                if (full && (g.N & 3) == 0) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int t = 0; t < 4 && n + t < g.N; ++t) p[t] = v[t];
                continue;
            }
            float o[4] = {v[0] * g.alpha, v[1] * g.alpha, v[2] * g.alpha, v[3] * g.alpha};
            if (g.accumulate) {  // C is f32 when accumulating (checked on the host)
                const float* cp = (const float*)g.C + (long long)m * g.ldc + n;
                if (full && g.c_vec) {
                    const float4 c = *reinterpret_cast<const float4*>(cp);
                    o[0] += c.x; o[1] += c.y; o[2] += c.z; o[3] += c.w;
                } else
                    for (int t = 0; t < 4 && n + t < g.N; ++t) o[t] += cp[t];
            }
            if (g.bias)
                for (int t = 0; t < 4 && n + t < g.N; ++t) o[t] += g.bias[n + t];
            if (g.act == EGK_ACT_RELU)
                for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f);
            if (g.residual) {
                if (g.r_bf16) {
                    const bf16_t* rp = (const bf16_t*)g.residual + (long long)m * g.ldr + n;
                    if (full && g.r_vec) {
                        const uint2 r = *reinterpret_cast<const uint2*>(rp);
                        o[0] += __uint_as_float(r.x << 16); o[1] += __uint_as_float(r.x & 0xffff0000u);
                        o[2] += __uint_as_float(r.y << 16); o[3] += __uint_as_float(r.y & 0xffff0000u);
                    } else
                        for (int t = 0; t < 4 && n + t < g.N; ++t) o[t] += bf16_to_f32(rp[t]);
                } else {
                    const float* rp = (const float*)g.residual + (long long)m * g.ldr + n;
                    if (full && g.r_vec) {
                        const float4 r = *reinterpret_cast<const float4*>(rp);
                        o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
                    } else
                        for (int t = 0; t < 4 && n + t < g.N; ++t) o[t] += rp[t];
                }
            }
            if (g.c_bf16) {
                bf16_t* cp = (bf16_t*)g.C + (long long)m * g.ldc + n;
                if (full && g.c_vec) {
                    uint2 pk;
                    pk.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
                    pk.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
                    *reinterpret_cast<uint2*>(cp) = pk;
                } else
                    for (int t = 0; t < 4 && n + t < g.N; ++t) cp[t] = f32_to_bf16(o[t]);
            } else {
                float* cp = (float*)g.C + (long long)m * g.ldc + n;
                if (full && g.c_vec) *reinterpret_cast<float4*>(cp) = make_float4(o[0], o[1], o[2], o[3]);
                else
                    for (int t = 0; t < 4 && n + t < g.N; ++t) cp[t] = o[t];
            }
        }
    }
}

// Epilogue of the 4-wave pipelined variants through LDS.  The accumulator layout gives a lane 4 consecutive columns of
// ONE row per MFMA tile: stored directly, a wave-instruction writes 16 rows x 32 bytes (bf16) -- 16 partial-line
// requests for 512 bytes -- and a 96 x 128 tile took 3.7 us just to ISSUE its stores (in-kernel stamps,
// tools/gemm_stamps.py --phases: 2 us to the first K tile, 9.4 us of K loop, 3.7 us of epilogue for 6144 x 1024 x
// 1024).  Here the raw accumulators go to the (now idle) ring as f32 [rows][128], 16-byte slot index XOR (row & 15)
// (conflict-free for the scattered writes and the row reads), and come back as 8 consecutive columns per lane: a
// wave-instruction then covers 4 whole 256-byte (bf16) rows.  The arithmetic per element is gemm_epilogue's, in its
// order: results are bit-identical.  Block-uniform precondition (epilogue_rows_ok): 16-byte aligned rows everywhere.
__host__ __device__ __forceinline__ bool epilogue_rows_ok(const GemmArgs& g) {
    if (!g.rows_epilogue) return false;
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((g.N & 7) != 0) return false;
    if (g.splitk > 1) return al16(g.ws);  // slabs: [z][M][N] f32, N % 8 == 0
    if (!al16(g.C) || (g.ldc & (g.c_bf16 ? 7 : 3)) != 0) return false;
    if (g.bias && !al16(g.bias)) return false;
    if (g.residual && (!al16(g.residual) || (g.ldr & (g.r_bf16 ? 7 : 3)) != 0)) return false;
    return true;
}

// MBW = 2: the 8-wave tall tiles (waves as 4 x 2 over 32 * NI * 2 rows): the same staging image with twice the rows, read out 32
// rows per pass by the 512 threads; the statistics of the 8 waves are summed in wave order, pairwise.
template <int NI, bool ST = true, int MBW = 1>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmArgs& g, const f32x4 (&acc)[NI][4], unsigned char* lds, int m0, int n0,
                                                   int wm, int wn, int lr, int lg, int z, int tid, int st_tile = 0) {
    __syncthreads();  // every wave is done with the ring (nothing is in flight: the last tiles were waited for)
    float* stage = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = wm * 16 * NI + i * 16 + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = (wn * 16 + j * 4 + lg) ^ (row & 15);
            *reinterpret_cast<f32x4*>(stage + row * 128 + slot * 4) = acc[i][j];
        }
    }
    __syncthreads();
    const int c8 = tid & 15, n = n0 + c8 * 8;
    // per-segment sums of the stored result for the graph LayerNorm that consumes it.  A tile spans at most two row segments
    // (checked on the host): s0 = segment of the tile's first row, rel = 0 / 1.
    const int st_mode = (ST && g.splitk == 1) ? g.st_mode : 0;  // (ST = false: no statistics and no static LDS in the kernel)
    float sa[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    int s0 = 0, s_split = 0x7fffffff;
    float st_mu[2] = {0.f, 0.f}, st_ri[2] = {0.f, 0.f}, lw[8], lb[8];
    if (st_mode) {
        while (s0 + 1 < g.st_nseg && m0 >= g.st_seg_ptr[s0 + 1]) ++s0;
        if (s0 + 1 < g.st_nseg) s_split = g.st_seg_ptr[s0 + 1];
        if (st_mode == 2 && n < g.N) {
            st_mu[0] = g.st_stats[s0 * 2]; st_ri[0] = g.st_stats[s0 * 2 + 1];
            if (s0 + 1 < g.st_nseg) { st_mu[1] = g.st_stats[s0 * 2 + 2]; st_ri[1] = g.st_stats[s0 * 2 + 3]; }
#pragma unroll
            for (int t = 0; t < 8; ++t) { lw[t] = g.st_w[n + t]; lb[t] = g.st_b[n + t]; }
        }
    }
    const bool st_cols = n < g.N;
    if (n >= g.N && !st_mode) return;  // (N % 8 == 0: a group of 8 columns is all in or all out)
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g.bias && g.splitk == 1) {
        const float4 b0 = *reinterpret_cast<const float4*>(g.bias + n), b1 = *reinterpret_cast<const float4*>(g.bias + n + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
#pragma unroll
    for (int it = 0; it < 2 * NI; ++it) {
        const int row = it * (16 * MBW) + (tid >> 4), m = m0 + row;
        if (m >= g.M || !st_cols) continue;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * 128 + (((2 * c8) ^ (row & 15)) << 2));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * 128 + (((2 * c8 + 1) ^ (row & 15)) << 2));
        float o[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (g.splitk > 1) {
            float* p = g.ws + ((long long)z * g.M + m) * g.N + n;
            *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
            continue;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] *= g.alpha;
        if (g.accumulate) {  // C is f32 when accumulating (checked on the host)
            const float* cp = (const float*)g.C + (long long)m * g.ldc + n;
            const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
            o[0] += c0.x; o[1] += c0.y; o[2] += c0.z; o[3] += c0.w; o[4] += c1.x; o[5] += c1.y; o[6] += c1.z; o[7] += c1.w;
        }
        if (g.bias)
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] += bias[t];
        if (g.act == EGK_ACT_RELU)
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] = fmaxf(o[t], 0.f);
        if (g.residual) {
            if (g.r_bf16) {
                const uint4 r = *reinterpret_cast<const uint4*>((const bf16_t*)g.residual + (long long)m * g.ldr + n);
                const unsigned rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    o[2 * t] += __uint_as_float(rw[t] << 16);
                    o[2 * t + 1] += __uint_as_float(rw[t] & 0xffff0000u);
                }
            } else {
                const float* rp = (const float*)g.residual + (long long)m * g.ldr + n;
                const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                o[0] += r0.x; o[1] += r0.y; o[2] += r0.z; o[3] += r0.w; o[4] += r1.x; o[5] += r1.y; o[6] += r1.z; o[7] += r1.w;
            }
        }
        if (g.c_bf16) {
            uint4 pk;
            pk.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
            pk.z = (unsigned)f32_to_bf16(o[4]) | ((unsigned)f32_to_bf16(o[5]) << 16);
            pk.w = (unsigned)f32_to_bf16(o[6]) | ((unsigned)f32_to_bf16(o[7]) << 16);
            *reinterpret_cast<uint4*>((bf16_t*)g.C + (long long)m * g.ldc + n) = pk;
            if (st_mode) {  // the sums are those of the values a reader of C would READ: the rounded ones
                const unsigned pw[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    o[2 * t] = __uint_as_float(pw[t] << 16);
                    o[2 * t + 1] = __uint_as_float(pw[t] & 0xffff0000u);
                }
            }
        } else {
            float* cp = (float*)g.C + (long long)m * g.ldc + n;
            *reinterpret_cast<float4*>(cp) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(cp + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (st_mode == 1) {
            const int rel = m >= s_split ? 1 : 0;
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                a1 += o[t];
                a2 += o[t] * o[t];
            }
            sa[rel][0] += a1;
            sa[rel][1] += a2;
        } else if (st_mode == 2) {
            const int rel = m >= s_split ? 1 : 0;
            const float mu = st_mu[rel], ri = st_ri[rel];
            float xv[8];
            if (g.c_bf16) {
                const uint4 r = *reinterpret_cast<const uint4*>((const bf16_t*)g.st_x + (long long)m * g.st_ldx + n);
                const unsigned rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    xv[2 * t] = __uint_as_float(rw[t] << 16);
                    xv[2 * t + 1] = __uint_as_float(rw[t] & 0xffff0000u);
                }
            } else {
                const float* xp = (const float*)g.st_x + (long long)m * g.st_ldx + n;
                const float4 x0 = *reinterpret_cast<const float4*>(xp), x1 = *reinterpret_cast<const float4*>(xp + 4);
                xv[0] = x0.x; xv[1] = x0.y; xv[2] = x0.z; xv[3] = x0.w; xv[4] = x1.x; xv[5] = x1.y; xv[6] = x1.z; xv[7] = x1.w;
            }
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {  // graphln_bwd_stats_kernel's arithmetic: dxhat = dy * lrelu'(pre) * w
                const float xh = (xv[t] - mu) * ri;
                const float pre = xh * lw[t] + lb[t];
                const float dxh = o[t] * (pre > 0.f ? 1.f : g.st_slope) * lw[t];
                a1 += dxh;
                a2 += dxh * xh;
            }
            sa[rel][0] += a1;
            sa[rel][1] += a2;
        }
    }
    if constexpr (ST) if (st_mode) {  // block sums in a fixed order: lanes (shuffle tree), then the 4 waves in wave order
        // (the 8-wave tile keeps the wave partials behind its staging image in the ring -- every byte of the CU's LDS may belong to
        //  the dynamic ring there (a 4-stage ring is 160 KiB): no static allocation)
        __shared__ double st_red_static[MBW == 1 ? 4 : 1][4];
        double (*st_red)[4] = st_red_static;
        if constexpr (MBW == 2) st_red = reinterpret_cast<double (*)[4]>(lds + 32 * NI * MBW * 512);
        double v4[4] = {(double)sa[0][0], (double)sa[0][1], (double)sa[1][0], (double)sa[1][1]};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o_ = 32; o_ > 0; o_ >>= 1) v4[q] += __shfl_xor(v4[q], o_, 64);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) st_red[tid >> 6][q] = v4[q];
        }
        __syncthreads();
        if (tid < g.st_nseg * 2) {
            const int sg = tid >> 1, k = tid & 1, rel = sg - s0;
            double t_ = 0.0;
            if (rel == 0 || rel == 1) {
                t_ = (st_red[0][rel * 2 + k] + st_red[1][rel * 2 + k]) + (st_red[2][rel * 2 + k] + st_red[3][rel * 2 + k]);
                if constexpr (MBW == 2)
                    t_ += (st_red[4][rel * 2 + k] + st_red[5][rel * 2 + k]) + (st_red[6][rel * 2 + k] + st_red[7][rel * 2 + k]);
            }
            g.st_ws[((long long)st_tile * g.st_nseg + sg) * 2 + k] = t_;
        }
    }
}

// BF16C: bf16 MFMA (else exact f32).  TA/TB: operand transposed in memory.  AT/BT: element type in memory.
template <bool BF16C, bool TA, bool TB, typename AT, typename BT>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BM * ROWB];
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + BM * ROWB;
    constexpr int KT = BF16C ? 64 : 32;
    constexpr bool A16 = sizeof(AT) == 2, B16 = sizeof(BT) == 2;

    // XCD-aware tile id: blocks b, b+8, ... share an XCD; give each XCD a contiguous id range.
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int nkt = total_tiles(g, KT);
    const int z = blockIdx.y;  // split-K: this block handles K-tiles [t_begin, t_end)
    const int per = (nkt + g.splitk - 1) / g.splitk;
    const int t_begin = z * per, t_end = min(nkt, t_begin + per);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int lr = lane & 15, lg = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[8], rb[8];
    auto load_tile = [&](int t) {
        int src, tt;
        source_of<KT>(g, t, src, tt);
        const int k0 = tt * KT;
        load_operand<BF16C, TA, AT>(g.A[src], g.lda[src], g.M, m0, k0, g.K[src], (g.a_vec >> src) & 1, 0, ra);
        load_operand<BF16C, TB, BT>(g.B[src], g.ldb[src], g.N, n0, k0, g.K[src], (g.b_vec >> src) & 1, 1, rb);
    };

    if (t_begin < t_end) load_tile(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();  // every wave finished reading the previous image
        store_lds<TA, A16>(ldsA, ra, 0);
        store_lds<TB, B16>(ldsB, rb, 1);
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);  // in flight under the MFMAs below
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const uint4*>(ldsA + lds_off(wm * 64 + i * 16 + lr, s * 4 + lg));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const uint4*>(ldsB + lds_off(wn * 64 + j * 16 + lr, s * 4 + lg));
            if constexpr (BF16C) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j]),
                                                                            __builtin_bit_cast(bf16x8, a[i]),
                                                                            acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float av = __uint_as_float(q == 0 ? a[i].x : q == 1 ? a[i].y : q == 2 ? a[i].z : a[i].w);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float bv =
                                __uint_as_float(q == 0 ? b[j].x : q == 1 ? b[j].y : q == 2 ? b[j].z : b[j].w);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    gemm_epilogue<4, 4>(g, acc, m0, n0, wm * 64, wn * 64, lr, lg, z);
}

// ---- pipelined contraction: bf16 operands in memory, K % 64 == 0 ---------------------------------------------
// LDS-DMA staging (global_load_lds, 16 B per lane, 1 KiB per wave-instruction: no VGPR staging, no ds_write) into
// an NSTAGE-deep ring; ONE raw s_barrier per K-tile and a COUNTED s_waitcnt vmcnt so that the loads of the next
// tiles stay in flight across the barrier (cdna guide T3/T4).  Two LDS images, 16 KiB per 128-row operand tile:
//   * row-major operand ([rows][K] in memory): image [128 rows][64 k] (128-B rows), fragments by ds_read_b128;
//     16-byte chunk index XOR (row>>1)&7;
//   * transposed operand ([K][rows] in memory: W for dX, dY / X for dW): image [64 k][128 rows] (256-B rows) kept
//     in MEMORY order and read with ds_read_b64_tr_b16, the hardware transposing read: a 16-lane group fetches a
//     4(k) x 16(rows) block and every lane receives the 4 k-values of ITS row -- two of them are the 8-element
//     MFMA fragment.  16-byte chunk index XOR 2*(k&3) + 8*((k>>3)&1): the 8 k-rows one half-wave touches land on
//     16 distinct slots of the bank row.
// The images are lane-linear (DMA), so each swizzle is applied on the SOURCE address a lane fetches and again on
// the read (guide rule 21).  Rows beyond M / N are clamped to valid memory: they only feed accumulators that are
// never stored.  The MFMA chain per accumulator is the one of the generic kernel: results are bit-identical.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// Per-wave cursor over the 4 pieces (1 KiB each) a wave fetches of one operand image per K tile.  The per-lane
// source addresses are computed ONCE; a K-tile step is a uniform pointer increment (the address arithmetic of 8
// DMA instructions per tile would otherwise cost as many VALU cycles as the tile's MFMAs at one wave per SIMD).
template <bool TR, int NP, int OFF = 0, int TOT = NP>
__device__ __forceinline__ void cursor_init(const bf16_t* (&pp)[TOT], long long& step, const bf16_t* __restrict__ base, long long ld,
                                            int rows_total, int row0, int k0, int first, int lane, int tiles_per_step) {
    // pieces first .. first + NP - 1 of an operand image made of 16-KiB sub-images of 128 rows (16 pieces each), kept in
    // slots OFF .. OFF + NP - 1 of the pointer array
    const bf16_t** p = pp + OFF;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = (first + i) & 15, sub_row0 = row0 + ((first + i) >> 4) * 128;
        if constexpr (TR) {  // piece = 4 k-rows of 256 B; lane -> (k = 4*piece + lane/16, slot = lane%16)
            const int k = piece * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (2 * (k & 3) + 8 * ((k >> 3) & 1));
            // a chunk is fetched whenever it lies inside the ALLOCATED row (ld): with a padded row stride the last
            // valid rows (e.g. 112..114 of 115) sit in a chunk that extends into the padding.  Chunks beyond the
            // row are redirected to the row's first chunk: they only feed output rows that are never stored.
            int col = sub_row0 + c * 8;
            if (col + 8 > ld) col = 0;
            p[i] = base + (long long)(k0 + k) * ld + col;
        } else {  // piece = 8 rows of 128 B; lane -> (row = 8*piece + lane/8, slot = lane%8)
            const int r = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            p[i] = base + (long long)min(sub_row0 + r, rows_total - 1) * ld + k0 + c * 8;
        }
    }
    step = (TR ? 64 * ld : 64) * tiles_per_step;
}
template <int NP, int OFF = 0, int TOT = NP>
__device__ __forceinline__ void cursor_issue(const bf16_t* (&pp)[TOT], long long step, unsigned char* img, int first) {
    const bf16_t** p = pp + OFF;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        __builtin_amdgcn_global_load_lds((glb_void_t*)p[i], (lds_void_t*)(img + (first + i) * 1024), 16, 0, 0);
        p[i] += step;
    }
}
template <bool TR, int NP>
struct OperandCursor {
    const bf16_t* p[NP];
    long long step;  // elements per cursor advance
    __device__ __forceinline__ void init(const bf16_t* __restrict__ base, long long ld, int rows_total, int row0, int k0,
                                         int first, int lane, int tiles_per_step) {
        cursor_init<TR, NP, 0, NP>(p, step, base, ld, rows_total, row0, k0, first, lane, tiles_per_step);
    }
    __device__ __forceinline__ void issue(unsigned char* img, int first) { cursor_issue<NP, 0, NP>(p, step, img, first); }
};

// (128 * MB) x 128 output tile, NSTAGE-deep ring of (A image | B image) = (16 * MB + 16) KiB per stage; every wave
// owns a 64 x 64 patch of the tile (4 x 4 MFMA accumulators).
// MB = 1, KG = 1: 4 waves as 2 x 2; two such workgroups share a CU and hide each other's latencies.
// MB = 1, KG = 2: 8 waves = two wave groups, each 2 x 2 over the SAME output tile with its own ring, walking alternate K
//         tiles; the partial accumulators meet through LDS at the end (each group finishes half of the rows).  For
//         launches that cannot put two workgroups on a CU: the K walk per wave halves and the two groups overlap
//         each other's DMA / LDS / MFMA phases, with no slab traffic and no second launch.
// MB = 2, KG = 1: 8 waves as 4 x 2 over a 256 x 128 tile, one workgroup per CU.  A CU takes in ~70 GB/s from L2
//         whatever the kernel does (MI355X_MICROARCH.md, gather-into-LDS table), so bytes fetched per flop bound the
//         rate: 48 KiB per K tile for 2 x the flops of the 32 KiB of a 128 x 128 tile.  For large outputs.
// F16: the 16-bit operands are IEEE half (v_mfma_f32_16x16x32_f16) instead of bf16 -- same images, same fragment reads, same
// accumulation; 11 significand bits instead of 8: the one-product screen of the nearest-prototype search (egk_topk_window) gets an
// error window eight times narrower.  Only the instantiations the host asks for (gemm_f16 below) exist.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4& b, const uint4& a, const f32x4& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}

template <int NSTAGE, bool TRA, bool TRB, int KG, int MB, int NI = 4, bool VIRT = false, bool F16 = false, int LW = 0>
__device__ __forceinline__ void gemm_pipe_body(const GemmArgs& g, const int bid) {
    // LW > 0: LW extra LOADER waves issue every LDS-DMA piece of the workgroup; the WG * KG compute waves only read fragments and
    // feed the matrix pipe.  A piece holds the wave that issues it until the CU's vector-memory path accepts it (100 - 185 cycles
    // inside a loaded phase): on the compute waves those cycles sit between a tile's barrier and its first MFMA.
    static_assert(KG == 1 || MB == 1, "wave groups and the tall tile are alternatives");
    // NI = MFMA row fragments per wave: 4 -> 128-row tiles; 3 -> 96-row tiles (row-major A only), for outputs whose
    // 128-row tiling leaves the CUs unevenly loaded (6144 x 1024: 384 tiles = 1.5 per CU; 512 tiles of 96 x 128 = 2)
    // MB = 2 with NI = 3: 192 x 128 tiles, 8 waves as 4 x 2 (48 x 64 each) -- 6144 x 1024 outputs are exactly 256 of them, one
    // per CU, at 40 KiB per K tile for 1.5 x the flops of the 128 x 128 tile's 32 KiB.  Measured (tools/round6/r192_bench.py): alone
    // it runs at the 96-row tiles' rate (20.9 against 19.6 us at K = 1024, 59.5 against 59.2 at K = 4608), in the step 1.340 against
    // 1.365 ms.  Built beside it and NOT kept: a 4-stage ring (every byte of LDS: 20.7 us, the step 1.345) and waves 4-7 staggered by
    // half a K tile against their SIMD partners (fragments held across the barrier: 23.7 us, 256 registers + spills, the step 1.45).
    static_assert(NI == 4 || ((NI == 3 || NI == 2) && !TRA && ((MB == 1 && (KG == 1 || NI == 2)) || (MB == 2 && KG == 1 && NI == 3))),
                  "96- / 64-row tiles: row-major A; two wave groups with 64-row tiles only; the tall 8-wave tile with NI = 3");
    constexpr int IMG = 16384, IMG_A = NI == 4 ? MB * IMG : MB * NI * 4096, STAGE = IMG_A + IMG;
    constexpr int NPB = 4 / MB;        // B pieces per wave per tile
    constexpr int LOADS = NI + NPB;    // wave-instructions per wave per tile
    constexpr int WG = 4 * MB;         // waves per wave group
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int KT = 64;
#ifdef EGK_GEMM_STAMPS  // diagnostic build only (tools/gemm_stamps.py --phases): entry, first tile landed, loop end, exit
    const unsigned long long ps_entry = __builtin_amdgcn_s_memtime(), ps_rentry = __builtin_amdgcn_s_memrealtime();
    unsigned long long ps_first = 0;
#endif

    int z, tm, tn;
    if constexpr (VIRT) tile_of_virtual(g, bid, z, tm, tn);  // (the caller did the XCD placement: grouped launches)
    else tile_of(g, bid, z, tm, tn);
    const int m0 = tm * (32 * NI * MB), n0 = tn * BN;
    const int nkt = total_tiles(g, KT);
    const int per = (nkt + g.splitk - 1) / g.splitk;
    const int t_begin = z * per, t_end = min(nkt, t_begin + per);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = KG == 1 ? 0 : (wall / WG);  // wave group (K-tile parity)
    const int w = wall % WG;
    const int wm = w >> 1, wn = w & 1;  // wm: 64-row band of the tile, 0 .. 2 * MB - 1
    const int lr = lane & 15, lg = lane >> 4;
    unsigned char* ring = lds + grp * (NSTAGE * STAGE);

    // this group's tiles: t_begin + grp + KG * i, i = 0 .. nt-1; every group runs ``rounds`` barrier rounds
    const int nt_all = t_end - t_begin;
    const int nt = nt_all > grp ? (nt_all - grp + KG - 1) / KG : 0;
    const int rounds = (nt_all + KG - 1) / KG;

    OperandCursor<TRA, NI> ca;
    OperandCursor<TRB, NPB> cb;
    int cur_src = -1;
    auto issue = [&](int i, int stage) {  // i-th tile of this group
        const int t = t_begin + grp + KG * i;
        int src, tt;
        source_of<KT>(g, t, src, tt);
        if (src != cur_src) {  // (uniform) first tile, or the walk crossed from one K source into the next
            const int k0 = tt * KT;
            ca.init((const bf16_t*)g.A[src], g.lda[src], g.M, m0, k0, w * NI, lane, KG);
            cb.init((const bf16_t*)g.B[src], g.ldb[src], g.N, n0, k0, w * NPB, lane, KG);
            cur_src = src;
        }
        unsigned char* sbase = ring + stage * STAGE;
        ca.issue(sbase, w * NI);
        cb.issue(sbase + IMG_A, w * NPB);
    };

    if constexpr (LW > 0) {
        static_assert(KG == 1 && (WG * NI) % LW == 0 && (WG * NPB) % LW == 0, "loader waves: one wave group, pieces divisible among the loaders");
        if (wall >= WG) {  // (wave-uniform) a loader wave: pieces [L * PA, (L + 1) * PA) of the A image, [L * PB, ..) of the B image
            constexpr int PA = WG * NI / LW, PB = WG * NPB / LW;
            const int L = wall - WG;
            OperandCursor<TRA, PA> la;
            OperandCursor<TRB, PB> lb;
            int lsrc = -1;
            auto lissue = [&](int i, int stage) {
                const int t = t_begin + i;
                int src, tt;
                source_of<KT>(g, t, src, tt);
                if (src != lsrc) {
                    const int k0 = tt * KT;
                    la.init((const bf16_t*)g.A[src], g.lda[src], g.M, m0, k0, L * PA, lane, 1);
                    lb.init((const bf16_t*)g.B[src], g.ldb[src], g.N, n0, k0, L * PB, lane, 1);
                    lsrc = src;
                }
                unsigned char* sbase = ring + stage * STAGE;
                la.issue(sbase, L * PA);
                lb.issue(sbase + IMG_A, L * PB);
            };
#pragma unroll
            for (int p = 0; p < NSTAGE - 1; ++p)
                if (p < nt) lissue(p, p);
            for (int it = 0; it < rounds; ++it) {
                const int later = min(NSTAGE - 2, nt - 1 - it);
                if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PA + PB)) : "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // tile ``it`` is in LDS for everybody; the stage read at it - 1 is free
                if (it + NSTAGE - 1 < nt) lissue(it + NSTAGE - 1, (it + NSTAGE - 1) % NSTAGE);
            }
            return;  // (the epilogue's barriers count the surviving waves only)
        }
    }

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-lane LDS byte offsets of the fragment reads (relative to the image base) -------------------------------
    const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)ring;
    // row-major image: (row, chunk) -> row*128 + ((chunk ^ ((row>>1)&7)) << 4); k-step toggles bit 6, fragment i adds i*2048
    const unsigned rm_sw = (unsigned)((lg ^ ((lr >> 1) & 7)) << 4);
    const unsigned a_rm = (unsigned)((wm * 16 * NI + lr) * ROWB) + rm_sw;  // (sub-images are contiguous: row r at r * 128)
    const unsigned b_rm = (unsigned)((wn * 64 + lr) * ROWB) + rm_sw;
    // k-major image: lane 4q+p of a 16-lane group addresses (k-row 8*lg + q [+4 for the 2nd half] [+32 per k-step],
    // 16-B chunk = (wave offset | fragment i | p>>1) ^ f, byte (p&1)*8), f = 2q + 8*(lg&1).  The fragment index enters
    // through an XOR, so each fragment has its own lane offset (4 per operand), k-step / half are immediates.
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned tr_row = (unsigned)((8 * lg + tq) * 256 + (tp & 1) * 8);
    const unsigned tr_f = (unsigned)(2 * tq + 8 * (lg & 1));
    unsigned a_tr[4], b_tr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_tr[i] = (unsigned)((wm >> 1) * IMG) + tr_row + ((((unsigned)((wm & 1) * 8 + 2 * i + (tp >> 1))) ^ tr_f) << 4);
        b_tr[i] = tr_row + ((((unsigned)(wn * 8 + 2 * i + (tp >> 1))) ^ tr_f) << 4);
    }

    // fused bias gradient (dW form: op(A) = dY^T, so sum_k op(A)[m, k] is the column sum of dY): the workgroups of
    // the first column tile also sum their k-major A image over k.  Thread -> 16-byte chunk column cc (8 output rows)
    // and a group of 4 k-rows; 8 f32 partial sums per thread across all K-tiles, combined through LDS at the end.
    const bool do_bias = TRA && g.dbias != nullptr && tn == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int bcc = tid & 15, bkg = (tid & 255) >> 4;
    const int bsub = MB == 2 ? (tid >> 8) : 0;  // tall tile: threads 256.. sum the second 128-row sub-image

    uint4 a0[4], a1[4], b0[4], b1[4];  // fragments of one K tile (a*[NI..3] stay unused for 96-row tiles)
    if constexpr (LW == 0) {
#pragma unroll
        for (int p = 0; p < NSTAGE - 1; ++p)
            if (p < nt) issue(p, p);
    }
    for (int it = 0; it < rounds; ++it) {
        // tile ``it`` has landed once this wave has at most the pieces of the LATER tiles already issued
        // (min(NSTAGE-2, nt-1-it) tiles x LOADS pieces) outstanding
        if constexpr (LW == 0) {
            const int later = min(NSTAGE - 2, nt - 1 - it);
            if (later >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LOADS) : "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // every wave's pieces of tile ``it`` landed; the stage read at it-1 is free
#ifdef EGK_GEMM_STAMPS
        if (it == 0) ps_first = __builtin_amdgcn_s_memtime();
#endif
        const unsigned stA = lds_base + (it % NSTAGE) * STAGE, stB = stA + IMG_A;
        if constexpr (LW == 0) {
            if (it + NSTAGE - 1 < nt) issue(it + NSTAGE - 1, (it + NSTAGE - 1) % NSTAGE);
        }
        if (KG > 1 && it >= nt) continue;  // (wave-group uniform) odd tile count: the last round is group 0's only

        // Fragment reads as inline asm: hipcc cannot prove that a plain ds_read does not alias the LDS-DMA writes
        // in flight and would put s_waitcnt vmcnt(0) in front of it, draining the prefetched tiles.
        uint4 bz[4];
        if constexpr (TRA) {
            if (do_bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = bkg * 4 + r;
                    const unsigned ad = stA + (unsigned)(bsub * IMG + k * 256 + ((bcc ^ (2 * (k & 3) + 8 * ((k >> 3) & 1))) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bz[r]) : "v"(ad));
                }
            }
        }
        if constexpr (TRA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ad = stA + a_tr[i];
                uint2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(ad));
                a0[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a0[i]) : "v"(stA + a_rm), "n"(i * 2048));
        }
        if constexpr (TRB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ad = stB + b_tr[j];
                uint2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(ad));
                b0[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b0[j]) : "v"(stB + b_rm), "n"(j * 2048));
        }
        if constexpr (TRA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ad = stA + a_tr[i];
                uint2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(lo) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(hi) : "v"(ad));
                a1[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a1[i]) : "v"((stA + a_rm) ^ 64u), "n"(i * 2048));
        }
        if constexpr (TRB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ad = stB + b_tr[j];
                uint2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(lo) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(hi) : "v"(ad));
                b1[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b1[j]) : "v"((stB + b_rm) ^ 64u), "n"(j * 2048));
        }
        // the reads return in issue order: k-step 0 is complete when only the k-step-1 reads are outstanding
        // (lgkmcnt is a 4-bit counter: with 16 k-step-1 reads, "<= 15 outstanding" already implies the first 16 are back)
        constexpr int R1 = ((TRA ? 8 : NI) + (TRB ? 8 : 4)) > 15 ? 15 : ((TRA ? 8 : NI) + (TRB ? 8 : 4));
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(R1) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = mfma16<F16>(b0[j], a0[i], acc[i][j]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = mfma16<F16>(b1[j], a1[i], acc[i][j]);
        if constexpr (TRA) {
            if (do_bias) {  // (the reads above completed at the lgkmcnt(0))
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned wv[4] = {bz[r].x, bz[r].y, bz[r].z, bz[r].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bsum[2 * e] += __uint_as_float(wv[e] << 16);
                        bsum[2 * e + 1] += __uint_as_float(wv[e] & 0xffff0000u);
                    }
                }
            }
        }
    }
#ifdef EGK_GEMM_STAMPS
    const unsigned long long ps_loop = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (TRA) {
        if (g.dbias != nullptr && tn == 0) {  // block-uniform
            __syncthreads();                  // every wave is done with the ring: reuse it as f32 scratch [16 * KG][128]
            float* red = reinterpret_cast<float*>(lds);
            const int slot = MB == 2 ? bsub : grp;  // [KG * MB slots][16 k groups][128 rows]
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(slot * 16 + bkg) * 128 + bcc * 8 + e] = bsum[e];
            __syncthreads();
            if (tid < 128 * MB && m0 + tid < g.M) {
                // MB = 1: the KG groups' partials of row tid; MB = 2: the 16 partials of sub-image tid / 128
                const int q0 = MB == 2 ? (tid >> 7) * 16 : 0, nq = MB == 2 ? 16 : 16 * KG;
                float t = 0.f;
                for (int q = 0; q < nq; ++q) t += red[(q0 + q) * 128 + (tid & 127)];
                if (g.splitk > 1) g.ws_bias[(long long)z * g.M + m0 + tid] = t;
                else g.dbias[m0 + tid] += t;
            }
            __syncthreads();
        }
    }
    if constexpr (KG == 1) {
        if constexpr (MB == 1) {
            if (epilogue_rows_ok(g)) gemm_epilogue_rows<NI>(g, acc, lds, m0, n0, wm, wn, lr, lg, z, tid, tm * g.tiles_n + tn);
            else gemm_epilogue<NI, 4>(g, acc, m0, n0, wm * 16 * NI, wn * 64, lr, lg, z);
        } else if constexpr (NI == 3) {  // 192 x 128: the staging image (96 KiB f32) fits the 3-stage ring
            static_assert(NSTAGE * STAGE >= 192 * 512 + 256, "the rows epilogue stages the whole tile (+ the wave partials) in the ring");
            if (epilogue_rows_ok(g)) gemm_epilogue_rows<NI, true, 2>(g, acc, lds, m0, n0, wm, wn, lr, lg, z, tid, tm * g.tiles_n + tn);
            else gemm_epilogue<NI, 4>(g, acc, m0, n0, wm * 16 * NI, wn * 64, lr, lg, z);
        } else {
            gemm_epilogue<NI, 4>(g, acc, m0, n0, wm * 16 * NI, wn * 64, lr, lg, z);
        }
    } else if constexpr (NI == 4 || NI == 2) {
        // Exchange: group 0 finishes the first NI / 2 row fragments of each wave tile, group 1 the others.  Each group parks
        // the half it does not finish in LDS ([group][wave][i2][j][lane] f32x4, lane-contiguous 16-B stores), then adds the
        // other group's half to its own (a + b in either order: the same bits in both groups).
        constexpr int HF = NI / 2;
        __syncthreads();  // all rings are dead
        f32x4* xch = reinterpret_cast<f32x4*>(lds);
        f32x4* mine_out = xch + ((grp * 4 + w) * (HF * 4)) * 64 + lane;
        const f32x4* theirs = xch + (((1 - grp) * 4 + w) * (HF * 4)) * 64 + lane;
        f32x4 half[HF][4];
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < HF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mine_out[(i * 4 + j) * 64] = acc[HF + i][j];
        } else {
#pragma unroll
            for (int i = 0; i < HF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mine_out[(i * 4 + j) * 64] = acc[i][j];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < HF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) half[i][j] = acc[i][j] + theirs[(i * 4 + j) * 64];
        } else {
#pragma unroll
            for (int i = 0; i < HF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) half[i][j] = theirs[(i * 4 + j) * 64] + acc[HF + i][j];
        }
        gemm_epilogue<HF, 4>(g, half, m0, n0, wm * 16 * NI + grp * 16 * HF, wn * 64, lr, lg, z);
    }
#ifdef EGK_GEMM_STAMPS
    if (g.ws_bias != nullptr && g.dbias == nullptr && tid == 0) {  // (the host points ws_bias behind the slabs in this build)
        const unsigned long long ps_issued = __builtin_amdgcn_s_memtime();  // the output stores are issued ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // ... and acknowledged
        unsigned long long* o = reinterpret_cast<unsigned long long*>(g.ws_bias) + (long long)bid * 8;
        const unsigned long long ps_end = __builtin_amdgcn_s_memtime();
        o[7] = ps_issued - ps_loop;
        o[0] = ps_first - ps_entry; o[1] = ps_loop - ps_first; o[2] = ps_end - ps_loop; o[3] = ps_end - ps_entry;
        o[4] = __builtin_amdgcn_s_memrealtime() - ps_rentry; o[5] = ps_rentry; o[6] = nt;
    }
#endif
}

// The slab sum of four consecutive columns (slab order: bitwise reproducible) + the epilogue: gemm_splitk_reduce's vector path, and
// the in-launch finish below.
__device__ __forceinline__ void splitk_finish_group(const GemmArgs& g, int m, int n, long long total) {
    const long long idx = (long long)m * g.N + n;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < g.splitk; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(g.ws + (long long)z * total + idx);
        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    float o[4] = {s4.x * g.alpha, s4.y * g.alpha, s4.z * g.alpha, s4.w * g.alpha};
    if (g.accumulate) {
        const float4 c = *reinterpret_cast<const float4*>((const float*)g.C + (long long)m * g.ldc + n);
        o[0] += c.x; o[1] += c.y; o[2] += c.z; o[3] += c.w;
    }
    if (g.bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] += g.bias[n + t];
    }
    if (g.act == EGK_ACT_RELU) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f);
    }
    if (g.residual) {
        if (g.r_bf16) {
            const uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)g.residual + (long long)m * g.ldr + n);
            o[0] += __uint_as_float(r.x << 16); o[1] += __uint_as_float(r.x & 0xffff0000u);
            o[2] += __uint_as_float(r.y << 16); o[3] += __uint_as_float(r.y & 0xffff0000u);
        } else {
            const float4 r = *reinterpret_cast<const float4*>((const float*)g.residual + (long long)m * g.ldr + n);
            o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
    }
    if (g.c_bf16) {
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
        pk.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
        *reinterpret_cast<uint2*>((bf16_t*)g.C + (long long)m * g.ldc + n) = pk;
    } else {
        *reinterpret_cast<float4*>((float*)g.C + (long long)m * g.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

template <int NSTAGE, bool TRA, bool TRB, int KG, int MB, int NI = 4, bool F16 = false, int LW = 0>
__global__ __launch_bounds__(NTHREADS * KG * MB + 64 * LW) void gemm_pipe_kernel(const GemmArgs g) {
    gemm_pipe_body<NSTAGE, TRA, TRB, KG, MB, NI, false, F16, LW>(g, blockIdx.x);
}

// Grouped launch: up to MAX_GROUPS independent contractions of the SAME layout / element types / tile variant in one
// launch (the projection heads of the task batches: same shapes, different rows of the backbone output, different
// weights).  blockIdx.y = problem, blockIdx.x = its tile id; gridDim.x is the largest tile count rounded up to a
// multiple of 8, so that blockIdx.x & 7 is still the XCD the hardware deals the workgroup to (linear id = y * gridDim.x
// + x) and every problem keeps its XCD-local tile order.  Workgroups beyond a problem's tile count leave at once.
// XCD-packed placement (``packed``, 1-D grid): the tiles of ALL problems form one list, problem after problem, and XCD x (linear
// workgroup id mod 8) takes a CONSECUTIVE run of it -- so an XCD works on one problem (at most two) at a time and its L2 sees
// that problem's operand strips once, instead of every XCD pulling the strips of every problem: six H x H weight gradients
// fetched 450 MB (8 XCDs x 6 problems x ~6 strips of 1.5 MB) for 151 MB of operands; packed, 8 x ~14 strips = ~170 MB.
constexpr int MAX_GROUPS = 8;
struct GemmGroup {
    GemmArgs p[MAX_GROUPS];
    int packed, count, total;  // packed placement: number of problems, total tile count
};
template <int NSTAGE, bool TRA, bool TRB, int KG, int MB, int NI = 4, bool F16 = false, int LW = 0>
__global__ __launch_bounds__(NTHREADS * KG * MB + 64 * LW) void gemm_pipe_group_kernel(const GemmGroup gg) {
    if (gg.packed) {
        const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
        const int q = gg.total >> 3, r = gg.total & 7;
        if (slot >= q + (xcd < r ? 1 : 0)) return;
        int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        int pi = 0;
        for (; pi + 1 < gg.count; ++pi) {  // (uniform scalar walk over <= 8 problems)
            const int t = gg.p[pi].tiles_m * gg.p[pi].tiles_n;
            if (v < t) break;
            v -= t;
        }
        gemm_pipe_body<NSTAGE, TRA, TRB, KG, MB, NI, true, F16, LW>(gg.p[pi], v);
        return;
    }
    const GemmArgs& g = gg.p[blockIdx.y];
    if ((int)blockIdx.x >= g.tiles_m * g.tiles_n * g.splitk) return;
    gemm_pipe_body<NSTAGE, TRA, TRB, KG, MB, NI, false, F16, LW>(g, blockIdx.x);
}


// Fragment reads of the 256 x 256 kernel: fragments FIRST .. FIRST + 3 of k-step S of one operand image (inline asm for
// the reason given in gemm_pipe_kernel: a plain LDS load would drain the DMA queue first).
template <bool TR, int FIRST, int S, int CNT = 4>
__device__ __forceinline__ void read_frags4(uint4 (&dst)[4], unsigned st, unsigned rm, const unsigned* tr) {
    if constexpr (TR) {
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const unsigned ad = st + tr[FIRST + i];
            uint2 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ad), "n"(S * 8192));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ad), "n"(S * 8192 + 1024));
            dst[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    } else {
        const unsigned ad = S ? ((st + rm) ^ 64u) : (st + rm);
#pragma unroll
        for (int i = 0; i < CNT; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[i]) : "v"(ad), "n"((FIRST + i) * 2048));
    }
}
template <int NI, int CNT = 4>
__device__ __forceinline__ void mfma_4x4(f32x4 (&acc)[NI][4], int i0, const uint4 (&a)[4], const uint4 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < CNT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a[i]),
                                                                     acc[i0 + i][j], 0, 0, 0);
}

// ---- 256 x 256 output tile: large outputs -------------------------------------------------------------------------
// 8 waves as 2 (rows) x 4 (columns), every wave a 128 x 64 patch (8 x 4 MFMA accumulators), ONE workgroup per CU (two
// waves per SIMD), 2-stage ring of (A image | B image), each image two 16-KiB sub-images of 128 rows: 64 KiB per stage,
// 128 KiB in all.  Same images, swizzles and MFMA chain per accumulator as gemm_pipe_kernel (results are bit-identical
// to it and to the generic kernel); what changes is the bytes per flop:
//   * through the CU's vector-memory path: 64 KiB per K tile for 8.4 MFLOP = 128 flop/B against 64 flop/B for a
//     128 x 128 tile.  That path takes 64 B/clk -- one 1-KiB DMA piece per 16 cycles -- and a wave that issues a piece
//     is held until it is accepted (~68 cycles per piece with four SIMDs feeding it): at 128 x 128 the pieces of a K
//     tile take the path as long as its MFMAs take the matrix pipes (512 cycles each), which is the ~65 % pipe
//     utilisation / ~70 GB/s per CU / ~1.1 PFLOP/s every 128-row variant ends at;
//   * from LDS: 12 fragment reads per 32 MFMAs per k-step instead of 8 per 16.
// In-kernel clock under this load: 1.5-1.7 GHz (tools/gemm_stamps.py), i.e. the matrix peak the loop can be held
// against is ~1.7 PFLOP/s, not 2.5.
// Only whole tiles (M, N multiples of 256) of outputs that fill the chip with them, and a long K walk (host policy); no
// fused bias gradient (dW outputs are far too small to come here).
// NI = MFMA row fragments per wave: 8 -> 256-row tiles; 6 -> 192 x 256 tiles (row-major A only), for outputs whose 256-row tiling
// leaves the last round half empty: 6144 x 4096 is 384 tiles of 256 rows (1.5 rounds of 256 CUs = two rounds of 256 rows) but 512
// tiles of 192 rows (two whole rounds of 192 rows): the A image then fills 24 of its 32 KiB.
template <bool TRA, bool TRB, int DMA_NBE = 8, int NI = 8>
__global__ __launch_bounds__(512) void gemm_big_kernel(const GemmArgs g) {
    static_assert(NI == 8 || (NI == 6 && !TRA), "192-row tiles: row-major A");
    constexpr int IMG = 16384, IMG_OP = 2 * IMG, STAGE = 2 * IMG_OP, KT = 64, HF = NI / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    int z, tm, tn;
    tile_of(g, blockIdx.x, z, tm, tn);
    const int m0 = tm * (32 * NI), n0 = tn * 256;
    const int nkt = total_tiles(g, KT);
    const int per = (nkt + g.splitk - 1) / g.splitk;
    const int t_begin = z * per, t_end = min(nkt, t_begin + per);
    const int nt = t_end - t_begin;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;  // wm: 128-row band (= A sub-image); wn: 64-column band (B sub-image wn >> 1)
    const int lr = lane & 15, lg = lane >> 4;

    // DMA roles.  A wave is held ~60-125 cycles per 1-KiB piece it issues (the CU's address path takes the pieces of all
    // four SIMDs): as long as the 64 MFMAs of its K tile take, when it issues an even share of 8.  Issued by everybody
    // behind the barrier, the matrix pipes idle meanwhile (in-kernel stamps, tools/gemm_stamps.py).  So the two waves of
    // a SIMD take turns: waves 0-3 ("early", one per SIMD) issue the A image and NBE of every 8 B pieces behind the
    // barrier while their partners 4-7 ("late") run MFMA phases; the late waves issue the remaining B pieces after their
    // second phase, under the early waves' MFMAs.
    // Addresses: the tile is whole (M, N multiples of 256: host policy), so piece q of an operand is a UNIFORM base plus a
    // per-lane offset that takes one of two values (the swizzle state alternates with q): 2 VGPRs per operand instead of a
    // 64-bit pointer per piece, and the K walk is scalar arithmetic.
    //   row-major [rows][K]: piece q = rows 8q .. 8q+7; lane -> row 8q + lane/8, chunk (lane%8) ^ (4*(q&1) | (lane/16)%4)
    //   k-major   [K][rows]: piece q = k-rows 4(q%16) .. +3 of sub-image q/16; lane -> k-row + lane/16,
    //                        chunk (lane%16) ^ (2*((lane/16)%4) + 8*x), x = ((q%16)/2)&1
    constexpr int NBE = DMA_NBE;  // B pieces per early wave (4 or 8); late waves take 8 - NBE each
    const bool early = w < 4;
    unsigned voffA[2], voffB[2];
    const char *baseA = nullptr, *baseB = nullptr;  // (uniform) tile origin of the current K tile
    long long lda_b = 0, ldb_b = 0;                 // row strides in bytes
    int cur_src = -1;
    auto lane_off = [&](bool tr, long long ld_bytes, int state) -> unsigned {
        if (tr) return (unsigned)((lane >> 4) * ld_bytes + ((((lane & 15) ^ (2 * ((lane >> 4) & 3) + 8 * state))) << 4));
        return (unsigned)((lane >> 3) * ld_bytes + (((lane & 7) ^ ((state << 2) | ((lane >> 4) & 3))) << 4));
    };
    auto aim = [&](int i) {  // (uniform) first tile, or the walk crossed from the first K source into the second
        const int t = t_begin + i;
        int src, tt;
        source_of<KT>(g, t, src, tt);
        if (src != cur_src) {
            const long long k0 = (long long)tt * KT;
            lda_b = g.lda[src] * 2; ldb_b = g.ldb[src] * 2;
            baseA = (const char*)g.A[src] + (TRA ? k0 * lda_b + (long long)m0 * 2 : (long long)m0 * lda_b + k0 * 2);
            baseB = (const char*)g.B[src] + (TRB ? k0 * ldb_b + (long long)n0 * 2 : (long long)n0 * ldb_b + k0 * 2);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                voffA[x] = lane_off(TRA, lda_b, x);
                voffB[x] = lane_off(TRB, ldb_b, x);
            }
            cur_src = src;
        }
    };
    // pieces first .. first + N - 1 of one operand (first a multiple of 4: the swizzle state of piece first + i depends
    // on i alone), then nothing else: the K advance is applied once per tile by the caller
    auto issue_pieces = [&](auto trc, auto nc, const char* base, long long ld_bytes, const unsigned (&voff)[2], int first,
                            unsigned char* img) {
        constexpr bool TR = decltype(trc)::value;
        constexpr int N = decltype(nc)::value;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int q = first + i;
            const long long off = TR ? (long long)(4 * (q & 15)) * ld_bytes + (q >> 4) * 256 : (long long)(8 * q) * ld_bytes;
            const unsigned v = voff[TR ? ((i >> 1) & 1) : (i & 1)];
            __builtin_amdgcn_global_load_lds((glb_void_t*)(base + off + v), (lds_void_t*)(img + q * 1024), 16, 0, 0);
        }
    };
    using std::integral_constant;
    auto advance = [&]() {
        baseA += TRA ? KT * lda_b : KT * 2;
        baseB += TRB ? KT * ldb_b : KT * 2;
    };
    auto issue_early = [&](unsigned char* stage) {
        issue_pieces(integral_constant<bool, TRA>{}, integral_constant<int, NI>{}, baseA, lda_b, voffA, w * NI, stage);
        issue_pieces(integral_constant<bool, TRB>{}, integral_constant<int, NBE>{}, baseB, ldb_b, voffB, w * NBE, stage + IMG_OP);
    };
    auto issue_late = [&](unsigned char* stage) {
        if constexpr (NBE < 8)
            issue_pieces(integral_constant<bool, TRB>{}, integral_constant<int, 8 - NBE>{}, baseB, ldb_b, voffB,
                         4 * NBE + (w - 4) * (8 - NBE), stage + IMG_OP);
    };

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)lds;
    const unsigned rm_sw = (unsigned)((lg ^ ((lr >> 1) & 7)) << 4);
    const unsigned a_rm = (unsigned)((wm * 16 * NI + lr) * ROWB) + rm_sw;  // (NI = 8: wm * IMG; the sub-images are contiguous)
    const unsigned b_rm = (unsigned)((wn >> 1) * IMG + ((wn & 1) * 64 + lr) * ROWB) + rm_sw;
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned tr_row = (unsigned)((8 * lg + tq) * 256 + (tp & 1) * 8);
    const unsigned tr_f = (unsigned)(2 * tq + 8 * (lg & 1));
    unsigned a_tr[NI], b_tr[4];
#pragma unroll
    for (int i = 0; i < NI; ++i) a_tr[i] = (unsigned)(wm * IMG) + tr_row + ((((unsigned)(2 * i + (tp >> 1))) ^ tr_f) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        b_tr[j] = (unsigned)((wn >> 1) * IMG) + tr_row + ((((unsigned)((wn & 1) * 8 + 2 * j + (tp >> 1))) ^ tr_f) << 4);

#ifdef EGK_GEMM_STAMPS  // diagnostic build only (tools/gemm_stamps.py): where a wave's cycles go, per loop section
    unsigned long long st_wait = 0, st_issue = 0, st_head = 0, st_body = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime(), st_rbegin = __builtin_amdgcn_s_memrealtime();
#define EGK_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define EGK_STAMP(var)
#endif
    if (nt > 0) {
        aim(0);
        if (early) issue_early(lds);
        else issue_late(lds);
        advance();
    }
    for (int it = 0; it < nt; ++it) {
        EGK_STAMP(s0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // 2-stage ring: only tile ``it`` is in flight at this point
        __builtin_amdgcn_s_barrier();
        EGK_STAMP(s1);
        const bool more = it + 1 < nt;
        unsigned char* nxt = lds + ((it + 1) & 1) * STAGE;
        if (more) aim(it + 1);
        if (more && early) issue_early(nxt);
        EGK_STAMP(s2);

        // Four phases of 16 MFMAs per K tile: (k-step 0 | 1) x (A fragments 0-3 | 4-7).  The fragments of the NEXT phase
        // are read while this one's MFMAs run, into the registers the previous phase released: 64 fragment registers live
        // (two B sets, two A halves) beside the 128 accumulators, so two waves per SIMD fit in the register file.
        const unsigned stA = lds_base + (it & 1) * STAGE, stB = stA + IMG_OP;
        uint4 b0[4], b1[4], alo[4], ahi[4];
        constexpr int NA = TRA ? 8 : HF, NB = TRB ? 8 : 4;  // read instructions per group of fragments (A: HF fragments, B: 4)
        read_frags4<TRB, 0, 0>(b0, stB, b_rm, b_tr);
        read_frags4<TRA, 0, 0, HF>(alo, stA, a_rm, a_tr);
        read_frags4<TRA, HF, 0, HF>(ahi, stA, a_rm, a_tr);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NA) : "memory");  // reads return in issue order: b0, alo are back
        __builtin_amdgcn_sched_barrier(0);
        EGK_STAMP(s3);
        mfma_4x4<NI, HF>(acc, 0, alo, b0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags4<TRB, 0, 1>(b1, stB, b_rm, b_tr);
        read_frags4<TRA, 0, 1, HF>(alo, stA, a_rm, a_tr);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NA + NB > 15 ? 15 : NA + NB) : "memory");  // ahi (k-step 0) is back
        __builtin_amdgcn_sched_barrier(0);
        mfma_4x4<NI, HF>(acc, HF, ahi, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (more && !early) issue_late(nxt);
        if (more) advance();
        read_frags4<TRA, HF, 1, HF>(ahi, stA, a_rm, a_tr);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NA) : "memory");  // b1, alo (k-step 1) are back
        __builtin_amdgcn_sched_barrier(0);
        mfma_4x4<NI, HF>(acc, 0, alo, b1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_4x4<NI, HF>(acc, HF, ahi, b1);
#ifdef EGK_GEMM_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        EGK_STAMP(s4);
        st_wait += s1 - s0; st_issue += s2 - s1; st_head += s3 - s2; st_body += s4 - s3;
#endif
    }
#ifdef EGK_GEMM_STAMPS
    if (g.ws_bias != nullptr && lane == 0) {  // (the host points ws_bias at the caller's workspace in this build)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(g.ws_bias) + ((long long)blockIdx.x * 8 + w) * 8;
        o[0] = st_wait; o[1] = st_issue; o[2] = st_head; o[3] = st_body;
        o[4] = __builtin_amdgcn_s_memtime() - st_begin; o[5] = __builtin_amdgcn_s_memrealtime() - st_rbegin; o[6] = nt;
    }
#endif
#undef EGK_STAMP
    gemm_epilogue<NI, 4>(g, acc, m0, n0, wm * 16 * NI, wn * 64, lr, lg, z);
}

// ---- exact-f32 pipelined contraction: f32 operands in memory, every K source a multiple of 32 -----------------------------
// The reference computes in f32; this is the kernel of the 'f32' compute mode (v_mfma_f32_16x16x4_f32: an exact, k-ordered fmaf
// chain at 1/16 of the bf16 matrix rate).  Same structure as gemm_pipe_kernel -- LDS-DMA staging (no VGPR staging, no
// ds_write), a 2-stage ring of (A image | B image), ONE raw s_barrier per K tile, XCD-local tile order, row-contiguous
// epilogue -- with a 32-deep K tile, so that an image is again 128 rows x 128 B (row-major operand) or 32 k-rows x 512 B
// (operand transposed in memory):
//   * row-major operand: image [128 rows][32 k] f32, 16-byte chunk index XOR (row >> 1) & 7, fragments by ds_read_b128: the
//     four floats of a lane's chunk feed four consecutive MFMAs (k = 16 s + 4 lg + q);
//   * transposed operand ([K][rows] in memory): the image keeps the memory order [32 k][128 rows]; a lane reads ITS (row, k)
//     element with ds_read_b32; 16-byte chunk index XOR ((k >> 2) & 3) << 2 puts the two k-rows a 32-lane half touches on
//     disjoint bank halves.
// The matrix pipe needs 128 MFMAs x 32 cycles = 4096 cycles per wave and K tile against 8 DMA pieces and 16-64 LDS reads: the
// loop is MFMA-issue bound, the next step's fragments are read while this step's MFMAs run.  Every accumulator sees the
// k-ordered chain of the register-staged generic kernel: results are bit-identical to it.
template <bool TR, int NP>
__device__ __forceinline__ void cursor32_init(const float* (&p)[NP], long long& step, const float* __restrict__ base, long long ld,
                                              int rows_total, int row0, int k0, int first, int lane) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = first + i;  // 0 .. 15 of the 16-KiB image
        if constexpr (TR) {           // piece = 2 k-rows of 512 B; lane -> (k = 2 piece + lane / 32, chunk slot = lane % 32)
            const int k = piece * 2 + (lane >> 5);
            const int c = (lane & 31) ^ (((k >> 2) & 3) << 2);
            int col = row0 + c * 4;
            if (col + 4 > ld) col = 0;  // (beyond the allocated row: feeds output rows that are never stored)
            p[i] = base + (long long)(k0 + k) * ld + col;
        } else {                      // piece = 8 rows of 128 B; lane -> (row = 8 piece + lane / 8, chunk slot = lane % 8)
            const int r = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            p[i] = base + (long long)min(row0 + r, rows_total - 1) * ld + k0 + c * 4;
        }
    }
    step = TR ? 32 * ld : 32;
}

// fragments of one k-step S (16 k values = four MFMA rounds q) of one operand image: f[i][q]
template <bool TR, int S_, int NF>
__device__ __forceinline__ void read_step32(float (&f)[NF][4], unsigned st, unsigned rm, const unsigned (&tr)[4]) {
    if constexpr (TR) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NF; ++i)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[i][q]) : "v"(st + tr[i]), "n"((16 * S_ + q) * 512));
    } else {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(S_ ? ((st + rm) ^ 64u) : (st + rm)), "n"(i * 2048));
            f[i][0] = v[0]; f[i][1] = v[1]; f[i][2] = v[2]; f[i][3] = v[3];
        }
    }
}

// NI = MFMA row fragments per wave: 4 -> 128-row tiles; 3 -> 96-row tiles (row-major A only) for outputs whose 128-row tiling
// loads the CUs unevenly -- two co-resident workgroups share a CU's matrix pipes, so a launch takes as long as the rows on its
// fullest CU: 6144 x 1024 is 384 tiles of 128 rows (2 x 128 rows on half of the CUs) but 512 tiles of 96 rows (2 x 96 everywhere).
template <bool TRA, bool TRB, int NI = 4, bool VIRT = false>
__device__ __forceinline__ void gemm_pipe_f32_body(const GemmArgs& g, const int bid) {
    static_assert(NI == 4 || (NI == 3 && !TRA), "96-row tiles: row-major A");
    constexpr int IMG = 16384, IMG_A = NI * 4096, STAGE = IMG_A + IMG, KT = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int z, tm, tn;
    if constexpr (VIRT) tile_of_virtual(g, bid, z, tm, tn);  // (the caller did the XCD placement: grouped launches)
    else tile_of(g, bid, z, tm, tn);
    const int m0 = tm * (32 * NI), n0 = tn * BN;
    const int nkt = total_tiles(g, KT);
    const int per = (nkt + g.splitk - 1) / g.splitk;
    const int t_begin = z * per, t_end = min(nkt, t_begin + per);
    const int nt = t_end - t_begin;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int lr = lane & 15, lg = lane >> 4;

    const float* pa[NI];
    const float* pb[4];
    long long sa = 0, sb = 0;
    int cur_src = -1;
    auto issue = [&](int i, int stage) {
        int src, tt;
        source_of<KT>(g, t_begin + i, src, tt);
        if (src != cur_src) {  // (uniform) first tile, or the walk crossed into the next K source
            cursor32_init<TRA, NI>(pa, sa, (const float*)g.A[src], g.lda[src], g.M, m0, tt * KT, w * NI, lane);
            cursor32_init<TRB, 4>(pb, sb, (const float*)g.B[src], g.ldb[src], g.N, n0, tt * KT, w * 4, lane);
            cur_src = src;
        }
        unsigned char* sbase = lds + stage * STAGE;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)pa[q], (lds_void_t*)(sbase + (w * NI + q) * 1024), 16, 0, 0);
            pa[q] += sa;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)pb[q], (lds_void_t*)(sbase + IMG_A + (w * 4 + q) * 1024), 16, 0, 0);
            pb[q] += sb;
        }
    };

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)lds;
    // row-major image: (row, chunk) -> row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); k-step s toggles bit 6, fragment i adds i * 2048
    const unsigned rm_sw = (unsigned)((lg ^ ((lr >> 1) & 7)) << 4);
    const unsigned a_rm = (unsigned)((wm * 16 * NI + lr) * ROWB) + rm_sw, b_rm = (unsigned)((wn * 64 + lr) * ROWB) + rm_sw;
    // k-major image: element (k, row) at k * 512 + (((row >> 2) ^ (((k >> 2) & 3) << 2)) << 4) + (row & 3) * 4 with
    // k = 16 s + 4 lg + q: the swizzle term is lg << 2 for every (s, q), which become an immediate (16 s + q) * 512
    unsigned a_tr[4], b_tr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_tr[i] = (unsigned)(4 * lg * 512 + ((((wm * 16 + i * 4 + (lr >> 2)) ^ (lg << 2))) << 4) + (lr & 3) * 4);
        b_tr[i] = (unsigned)(4 * lg * 512 + ((((wn * 16 + i * 4 + (lr >> 2)) ^ (lg << 2))) << 4) + (lr & 3) * 4);
    }

    auto mfma_step = [&](const float (&fa)[NI][4], const float (&fb)[4][4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][q], fa[i][q], acc[i][j], 0, 0, 0);
    };
    // fused bias gradient (dW form: op(A) = dY^T, so sum_k op(A)[m, k] is the column sum of dY), as in the bf16 kernel: the
    // workgroups of the first column tile also sum their k-major A image over k.  Thread -> 16-byte chunk column bcc (4
    // output rows) and a group of 4 of the tile's 32 k-rows; combined through LDS at the end.
    const bool do_bias = TRA && g.dbias != nullptr && tn == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const int bcc = tid & 31, bkg = tid >> 5;
    if (nt > 0) issue(0, 0);
    for (int it = 0; it < nt; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // 2-stage ring: only tile ``it`` is in flight here
        __builtin_amdgcn_s_barrier();                    // every wave's pieces landed; the stage read at it - 1 is free
        if (it + 1 < nt) issue(it + 1, (it + 1) & 1);    // in flight under this tile's 4096 MFMA cycles
        const unsigned stA = lds_base + (it & 1) * STAGE, stB = stA + IMG_A;
        float a0[NI][4], b0[4][4], a1[NI][4], b1[4][4];
        f32x4 bz[4];
        if constexpr (TRA) {
            if (do_bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = bkg * 4 + r;
                    const unsigned ad = stA + (unsigned)(k * 512 + ((bcc ^ (((k >> 2) & 3) << 2)) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bz[r]) : "v"(ad));
                }
            }
        }
        read_step32<TRA, 0, NI>(a0, stA, a_rm, a_tr);
        read_step32<TRB, 0, 4>(b0, stB, b_rm, b_tr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        read_step32<TRA, 1, NI>(a1, stA, a_rm, a_tr);  // (in flight while the first step's MFMAs run)
        read_step32<TRB, 1, 4>(b1, stB, b_rm, b_tr);
        mfma_step(a0, b0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(a1, b1);
        if constexpr (TRA) {
            if (do_bias) {  // (the reads completed at the first lgkmcnt(0) of this tile)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsum[e] += bz[r][e];
            }
        }
    }
    if constexpr (TRA) {
        if (g.dbias != nullptr && tn == 0) {  // block-uniform
            __syncthreads();                  // every wave is done with the ring: reuse it as f32 scratch [8 k groups][128 rows]
            float* red = reinterpret_cast<float*>(lds);
#pragma unroll
            for (int e = 0; e < 4; ++e) red[bkg * 128 + bcc * 4 + e] = bsum[e];
            __syncthreads();
            if (tid < 128 && m0 + tid < g.M) {
                float t = 0.f;
                for (int q = 0; q < 8; ++q) t += red[q * 128 + tid];
                if (g.splitk > 1) g.ws_bias[(long long)z * g.M + m0 + tid] = t;
                else g.dbias[m0 + tid] += t;
            }
            __syncthreads();
        }
    }
    if (epilogue_rows_ok(g)) gemm_epilogue_rows<NI>(g, acc, lds, m0, n0, wm, wn, lr, lg, z, tid, tm * g.tiles_n + tn);
    else gemm_epilogue<NI, 4>(g, acc, m0, n0, wm * 16 * NI, wn * 64, lr, lg, z);
}

template <bool TRA, bool TRB, int NI = 4>
__global__ __launch_bounds__(NTHREADS) void gemm_pipe_f32_kernel(const GemmArgs g) {
    gemm_pipe_f32_body<TRA, TRB, NI, false>(g, blockIdx.x);
}

// grouped launch of exact-f32 contractions (the weight gradients of the reference-precision step): XCD-packed placement
// as gemm_pipe_group_kernel -- the tiles of all problems form one list, XCD x takes a consecutive run of it
template <bool TRA, bool TRB>
__global__ __launch_bounds__(NTHREADS) void gemm_pipe_f32_group_kernel(const GemmGroup gg) {
    const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
    const int q = gg.total >> 3, r = gg.total & 7;
    if (slot >= q + (xcd < r ? 1 : 0)) return;
    int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    int pi = 0;
    for (; pi + 1 < gg.count; ++pi) {  // (uniform scalar walk over <= 8 problems)
        const int t = gg.p[pi].tiles_m * gg.p[pi].tiles_n;
        if (v < t) break;
        v -= t;
    }
    gemm_pipe_f32_body<TRA, TRB, 4, true>(gg.p[pi], v);
}

// Sum the split-K slabs in slab order (bitwise reproducible) and apply the epilogue.
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const GemmArgs g) {
    const long long total = (long long)g.M * g.N;
    // four consecutive columns per thread, 16-byte slab loads (all slabs of an element group in flight together);
    // rows are walked with N / 4 groups each, so no element group straddles a row
    const bool vec4 = (g.N & 3) == 0 && g.c_vec && (!g.residual || g.r_vec);
    if (vec4) {
        const int gpr = g.N >> 2;  // groups per row
        const long long groups = (long long)g.M * gpr;
        for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < groups; q += (long long)gridDim.x * blockDim.x) {
            const int m = (int)(q / gpr), n = (int)(q - (long long)m * gpr) * 4;
            splitk_finish_group(g, m, n, total);
        }
    } else {
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
             idx += (long long)gridDim.x * blockDim.x) {
            const int m = (int)(idx / g.N), n = (int)(idx % g.N);
            float s = 0.f;
            for (int z = 0; z < g.splitk; ++z) s += g.ws[(long long)z * total + idx];
            float o = s * g.alpha;
            if (g.accumulate) o += ((const float*)g.C)[(long long)m * g.ldc + n];
            if (g.bias) o += g.bias[n];
            if (g.act == EGK_ACT_RELU) o = fmaxf(o, 0.f);
            if (g.residual)
                o += g.r_bf16 ? bf16_to_f32(((const bf16_t*)g.residual)[(long long)m * g.ldr + n])
                              : ((const float*)g.residual)[(long long)m * g.ldr + n];
            if (g.c_bf16) ((bf16_t*)g.C)[(long long)m * g.ldc + n] = f32_to_bf16(o);
            else ((float*)g.C)[(long long)m * g.ldc + n] = o;
        }
    }
    if (g.dbias && g.ws_bias)
        for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < g.M; m += (long long)gridDim.x * blockDim.x) {
            float t = 0.f;
            for (int z = 0; z < g.splitk; ++z) t += g.ws_bias[(long long)z * g.M + m];
            g.dbias[m] += t;
        }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool BF16C, typename AT, typename BT>
static void launch_layout(const egk_gemm_desc* d, dim3 grid, hipStream_t s, const GemmArgs& g) {
    dim3 block(NTHREADS);
    if (!d->transA && !d->transB) hipLaunchKernelGGL((gemm_kernel<BF16C, false, false, AT, BT>), grid, block, 0, s, g);
    else if (!d->transA && d->transB) hipLaunchKernelGGL((gemm_kernel<BF16C, false, true, AT, BT>), grid, block, 0, s, g);
    else if (d->transA && d->transB) hipLaunchKernelGGL((gemm_kernel<BF16C, true, true, AT, BT>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<BF16C, true, false, AT, BT>), grid, block, 0, s, g);
}

}  // namespace egk

using namespace egk;

static int g_use_pipe = 1;
// fixed cost of the reduce launch behind a split-K contraction in the bf16 cost model below, in microseconds.  3.5 is the
// stand-alone fit; inside a captured step every extra launch also pays a boundary on its queue (a one-lane kernel lasts ~4.7 us
// in a replay) -- development knob egk_gemm_set_pipeline(400 + tenths of a microsecond).
static double g_reduce_fixed_us = 3.5;
static int g_group_tt_pad_kb = 0;   // development knob (egk_gemm_set_pipeline(600 + KiB)): extra dynamic LDS of queued weight-gradient groups
static int g_wg2_rows64 = 0;        // development knob (egk_gemm_set_pipeline(500 / 501)): variant 12 inside the policy off / on
static int g_group_m_override = 0;  // development knob (egk_gemm_set_pipeline(100 + group_m); 100 = policy)
static int g_rows_epilogue = 1;     // development knob (egk_gemm_set_pipeline(200 / 201): direct / row-contiguous epilogue)
static int g_group_packed = 1;      // development knob (egk_gemm_set_pipeline(300 / 301): spread / XCD-packed placement of grouped launches)
static int g_row_affinity = 1;      // development knob (egk_gemm_set_pipeline(950 / 951)): XCD x owns a contiguous eighth of the tile rows off / on
static int g_group_r192 = 1;        // development knob (egk_gemm_set_pipeline(860 / 861)): the 192 x 128 tile for one-round row-major-A groups off / on
static int g_r192_loaders = 1;      // development knob (egk_gemm_set_pipeline(870 / 871)): the 192 x 128 tile with four loader waves (variant 19) off / on
static int g_tt_tall = 1;           // development knob (egk_gemm_set_pipeline(850 / 851)): 256 x 128 tiles for weight-gradient groups that leave the second workgroup slot of many CUs empty, off / on
static int g_r192 = 1;              // development knob (egk_gemm_set_pipeline(900 / 901)): 192 x 128 tiles (variant 16) inside the policy off / on
static bool g_lds_attr_set = false;
template <int NS, bool TA, bool TB, int KG, int MB = 1>
static void set_lds_attr() {
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<NS, TA, TB, KG, MB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              NS * KG * (MB + 1) * 16384);
}
static void ensure_lds_attr() {
    if (g_lds_attr_set) return;
    set_lds_attr<2, false, false, 1>(); set_lds_attr<2, false, true, 1>(); set_lds_attr<2, true, true, 1>(); set_lds_attr<2, true, false, 1>();
    set_lds_attr<2, false, false, 2>(); set_lds_attr<2, false, true, 2>(); set_lds_attr<2, true, true, 2>(); set_lds_attr<2, true, false, 2>();
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<2, false, false, 1, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<2, false, true, 1, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<2, false, false, 1, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<2, false, true, 1, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_big_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_big_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_big_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_big_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_big_kernel<false, false, 8, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_big_kernel<false, true, 8, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, false, 1, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, true, 1, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, false, 1, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, true, 1, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, false, 1, 1, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, true, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, true, true, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, true, true, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<3, true, true, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<3, false, false, 1, 2, 3, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<3, false, true, 1, 2, 3, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, false, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_group_kernel<2, false, true, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_group_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_group_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_group_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_kernel<false, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_f32_kernel<false, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<3, false, false, 1, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<3, false, true, 1, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<3, false, false, 1, 2, 3, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<3, false, true, 1, 2, 3, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<2, false, false, 2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
    (void)hipFuncSetAttribute((const void*)egk::gemm_pipe_kernel<2, false, true, 2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
    g_lds_attr_set = true;
}
// development knob (A/B runs in one process): 0 routes every contraction through the generic kernel
extern "C" int egk_gemm_set_pipeline(int32_t on) {
    const int prev = g_use_pipe;
    if (on >= 950) { g_row_affinity = on - 950; return prev; }
    if (on >= 900) { g_r192 = on - 900; return prev; }
    if (on >= 870) { g_r192_loaders = on - 870; return prev; }
    if (on >= 860) { g_group_r192 = on - 860; return prev; }
    if (on >= 850) { g_tt_tall = on - 850; return prev; }
    if (on >= 700) return prev;  // (700 / 80x: knobs of variants that no longer exist)
    if (on >= 600) { g_group_tt_pad_kb = on - 600; return prev; }
    if (on >= 500) { g_wg2_rows64 = on - 500; return prev; }
    if (on >= 400) { g_reduce_fixed_us = (on - 400) * 0.1; return prev; }
    if (on >= 300) { g_group_packed = on - 300; return prev; }
    if (on >= 200) { g_rows_epilogue = on - 200; return prev; }
    if (on >= 100) { g_group_m_override = on - 100; return prev; }
    // 0 generic kernel only; 1 default policy; 2 always 3-stage; 3 always 2-stage; 4 always 4-stage (all 128 x 128,
    // one wave group); 5 always two wave groups; 6 always the 256 x 128 tile (2-stage); 7 always the 256 x 256 tile (no fused
    // bias gradient: falls back to 3 with one); 8 / 11 the 96 x 128 / 64 x 128 tile where legal (row-major A); 16 the 192 x 128
    // tile (8 waves, 3-stage ring) where legal (row-major A, no split-K); 900 / 901: variant 16 inside the policy off / on;
    // 950 / 951: XCD x owns a contiguous eighth of the tile rows (row-major A) off / on
    g_use_pipe = on;
    return prev;
}

// Split-K policy (host side, deterministic).  A launch of at most 128 tiles leaves half of the 256 CUs idle and
// walks K serially; slabs shorten the walk at the price of a second launch that sums them.  Cost model in
// microseconds, fitted on MI355X to tools/gemm_bench.py (device time inside a hipGraph):
//   walk(s)   = 6.5 + 0.45 * ceil(K tiles / s)                  launch + prologue + epilogue, then per 64-deep K tile
//   reduce(s) = 3.5 + 1.5 * s * (M * N / 2^20)                  slabs written and read back at ~5.5 TB/s
// s = the power of two <= 16 with tiles * s <= 256 and at least 2 K tiles per slab that minimises walk + reduce.
extern "C" int egk_gemm_splitk(int32_t M, int32_t N, int32_t K, int32_t compute) {
    const int KT = compute == EGK_COMPUTE_BF16 ? 64 : 32;
    const int tiles = cdiv(M, BM) * cdiv(N, BN);
    const int nkt = cdiv(K, KT);
    if (compute != EGK_COMPUTE_BF16) {
        // exact-f32 contractions are bound by the matrix pipes (1/16 of the bf16 rate), which the co-resident workgroups of a CU
        // share: a launch lasts (workgroups on the fullest CU) x (one workgroup's K walk), so what slabs buy is an even load --
        // 288 tiles keep 32 CUs busy twice as long as the rest, 288 x 4 slabs load every CU within 10 %.  Cost in microseconds:
        //   walk(s)   = ceil(tiles * s / 256) * ceil(nkt / s) * 1.9        (4096 MFMA cycles per 32-deep K tile at ~2.2 GHz)
        //   reduce(s) = 3.5 + 1.5 * s * (M * N / 2^20)                      (slabs written and read back)
        // (rows on the fullest CU in units of a 128-row tile: 96-row tiles are the kernel's alternative for row-major A)
        if (nkt < 8) return 1;
        const double mn = (double)M * N / 1048576.0;
        const long long t96 = (long long)cdiv(M, 96) * cdiv(N, BN);
        auto rounds = [&](int s_) {
            const double r128 = (double)cdiv((long long)tiles * s_, 256), r96 = 0.75 * cdiv(t96 * s_, 256);
            return r128 < r96 ? r128 : r96;
        };
        int best = 1;
        double best_cost = rounds(1) * nkt * 1.9;
        for (int s = 2; s <= 16 && nkt / s >= 4; ++s) {
            const double cost = rounds(s) * cdiv(nkt, s) * 1.9 + 3.5 + 1.5 * s * mn;
            if (cost < 0.93 * best_cost) { best_cost = cost; best = s; }
        }
        return best;
    }
    if (tiles > 128 || nkt < 4) return 1;
    const double mn = (double)M * N / 1048576.0;
    int best = 1;
    double best_cost = 6.5 + 0.45 * nkt;
    for (int s = 2; s <= 16 && tiles * s <= 256 && nkt / s >= 2; s *= 2) {
        const double cost = 6.5 + 0.45 * cdiv(nkt, s) + g_reduce_fixed_us + 1.5 * s * mn;
        if (cost < best_cost - 0.5) { best_cost = cost; best = s; }
    }
    return best;
}

extern "C" int64_t egk_gemm_ws_bytes(const egk_gemm_desc* d) {
    if (!d) return 0;
    const int sk = d->splitk > 1 ? d->splitk : 1;
    int64_t n = sk > 1 ? (int64_t)sk * d->M * d->N * 4 : 0;
    if (d->dbias) {
        const int64_t fused = sk > 1 ? (int64_t)sk * d->M * 4 : 0;
        const int64_t plain = (int64_t)egk_colsum_ws_len(d->K1, d->M) * 4;
        n += fused > plain ? fused : plain;
    }
    return n;
}

// The K sources of a descriptor -- (A1, B1, K1), (A2, B2, K2), then the n_extra extra ones, empty ones skipped -- into g.
// ea / eb: elements per 16 bytes of the operand element types.  Returns 0 or an error code (message set).
struct SourceInfo {
    long long k_total = 0;
    bool all_k64 = true, all_vec = true;
};
static int fill_sources(const egk_gemm_desc* d, GemmArgs& g, int ea, int eb, SourceInfo& info, const char* who) {
    EGK_REQUIRE(d->n_extra >= 0 && d->n_extra <= MAXSRC - 2, "%s: n_extra must be 0 .. %d", who, MAXSRC - 2);
    g.nsrc = 0;
    g.a_vec = g.b_vec = 0;
    for (int i = 0; i < MAXSRC; ++i) { g.K[i] = 0; g.A[i] = g.B[i] = nullptr; g.lda[i] = g.ldb[i] = 0; }
    auto add = [&](const void* A, const void* B, long long lda, long long ldb, int K) -> int {
        EGK_REQUIRE(K >= 0, "%s: negative K", who);
        if (K == 0) return 0;
        EGK_REQUIRE(A && B, "%s: null operand of a non-empty K source", who);
        const int i = g.nsrc++;
        g.K[i] = K; g.A[i] = A; g.B[i] = B; g.lda[i] = lda; g.ldb[i] = ldb;
        const bool av = aligned16(A) && (lda % ea == 0), bv = aligned16(B) && (ldb % eb == 0);
        g.a_vec |= (av ? 1 : 0) << i;
        g.b_vec |= (bv ? 1 : 0) << i;
        info.k_total += K;
        info.all_k64 = info.all_k64 && (K % 64 == 0);
        info.all_vec = info.all_vec && av && bv;
        return 0;
    };
    int rc = add(d->A1, d->B1, d->lda1, d->ldb1, d->K1);
    if (rc) return rc;
    rc = add(d->A2, d->B2, d->lda2, d->ldb2, d->K2);
    if (rc) return rc;
    for (int i = 0; i < d->n_extra; ++i) {
        rc = add(d->xA[i], d->xB[i], d->xlda[i], d->xldb[i], d->xK[i]);
        if (rc) return rc;
    }
    if (g.nsrc == 0) g.nsrc = 1;  // (K = 0: one empty source, the kernels store epilogue(0))
    return 0;
}

static int gemm_core(egk_stream_t stream, const egk_gemm_desc* d, int* query_blocks);

extern "C" int egk_gemm(egk_stream_t stream, const egk_gemm_desc* d) { return gemm_core(stream, d, nullptr); }

// Number of per-tile partial blocks [blocks][st_nseg][2] a launch of ``d`` with st_mode != 0 writes to st_ws -- 0 when the
// tile variant the policy picks for ``d`` cannot (the caller then runs the LayerNorm's own statistics pass).  Nothing is
// launched.
extern "C" int egk_gemm_stats_blocks(const egk_gemm_desc* d) {
    int blocks = 0;
    const int rc = gemm_core(nullptr, d, &blocks);
    return rc == 0 ? blocks : 0;
}

static int gemm_core(egk_stream_t stream, const egk_gemm_desc* d, int* query_blocks) {
    EGK_REQUIRE(d != nullptr, "egk_gemm: null descriptor");
    EGK_REQUIRE(d->M >= 0 && d->N >= 0 && d->K1 >= 0 && d->K2 >= 0, "egk_gemm: negative size");
    EGK_REQUIRE(d->compute == EGK_COMPUTE_F32 || d->compute == EGK_COMPUTE_BF16, "egk_gemm: bad compute type");
    EGK_REQUIRE(!d->op_f16, "egk_gemm: op_f16 is a grouped-launch feature (egk_gemm_grouped with >= 2 problems)");
    const bool a16 = d->a_dtype == EGK_BF16, b16 = d->b_dtype == EGK_BF16;
    EGK_REQUIRE((d->a_dtype == EGK_F32 || a16) && (d->b_dtype == EGK_F32 || b16) &&
                    (d->c_dtype == EGK_F32 || d->c_dtype == EGK_BF16) &&
                    (d->r_dtype == EGK_F32 || d->r_dtype == EGK_BF16),
                "egk_gemm: unknown element type");
    EGK_REQUIRE(a16 == b16, "egk_gemm: A and B must have the same element type in memory");
    EGK_REQUIRE(!(a16 && d->compute == EGK_COMPUTE_F32), "egk_gemm: bf16 operands need EGK_COMPUTE_BF16");
    EGK_REQUIRE(!(d->accumulate && d->c_dtype != EGK_F32), "egk_gemm: accumulate needs an f32 C");
    EGK_REQUIRE(d->C != nullptr || d->M == 0 || d->N == 0, "egk_gemm: null C");
    if (d->M == 0 || d->N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;

    GemmArgs g;
    g.M = d->M; g.N = d->N;
    const int ea = a16 ? 8 : 4, eb = b16 ? 8 : 4;  // elements per 16 bytes
    SourceInfo src;
    {
        const int rc = fill_sources(d, g, ea, eb, src, "egk_gemm");
        if (rc) return rc;
    }
    g.C = d->C; g.ldc = d->ldc;
    g.c_bf16 = d->c_dtype == EGK_BF16;
    g.c_vec = g.c_bf16 ? ((reinterpret_cast<uintptr_t>(g.C) & 7) == 0 && g.ldc % 4 == 0) : (aligned16(g.C) && g.ldc % 4 == 0);
    g.accumulate = d->accumulate; g.act = d->act; g.alpha = d->alpha;
    g.bias = d->bias; g.residual = d->residual; g.ldr = d->ldr;
    g.r_bf16 = d->r_dtype == EGK_BF16;
    g.r_vec = g.residual && (g.r_bf16 ? ((reinterpret_cast<uintptr_t>(g.residual) & 7) == 0 && g.ldr % 4 == 0)
                                      : (aligned16(g.residual) && g.ldr % 4 == 0));
    g.splitk = d->splitk > 1 ? d->splitk : 1;
    g.ws = (float*)d->ws;
    if (g.splitk > 1)
        EGK_REQUIRE(g.ws && d->ws_bytes >= (int64_t)g.splitk * d->M * d->N * 4, "egk_gemm: split-K workspace too small");
    g.tiles_m = cdiv(g.M, BM); g.tiles_n = cdiv(g.N, BN);
    g.dbias = nullptr; g.ws_bias = nullptr;
    g.group_m = 1;
    g.rows_epilogue = g_rows_epilogue;
    g.st_mode = d->st_mode; g.st_nseg = d->st_nseg; g.st_seg_ptr = d->st_seg_ptr; g.st_ws = (double*)d->st_ws;
    g.st_x = d->st_x; g.st_ldx = d->st_ldx; g.st_stats = d->st_stats; g.st_w = d->st_w; g.st_b = d->st_b; g.st_slope = d->st_slope;
    if (d->st_mode) {
        EGK_REQUIRE(d->st_mode == 1 || d->st_mode == 2, "egk_gemm: st_mode must be 0, 1 or 2");
        EGK_REQUIRE(d->st_nseg >= 1 && d->st_nseg <= 16 && d->st_seg_ptr && (query_blocks || d->st_ws), "egk_gemm: st_* segments / workspace");
        EGK_REQUIRE(d->st_mode == 1 || (d->st_x && d->st_stats && d->st_w && d->st_b), "egk_gemm: st_mode 2 needs x, stats, w, b");
    }
    if (query_blocks) *query_blocks = 0;
    const long long K = src.k_total;
    const double flops = 2.0 * d->M * d->N * K;
    const double bytes = (a16 ? 2.0 : 4.0) * ((double)d->M * K + (double)d->N * K) + (g.c_bf16 ? 2.0 : 4.0) * d->M * d->N;
    const int bf = d->compute == EGK_COMPUTE_BF16;
    const int layout = d->transA ? (d->transB ? 2 : 3) : (d->transB ? 1 : 0);  // nn, nt, tt, tn
    const double slab_bytes = d->splitk > 1 ? ((double)d->splitk + 1.0) * d->M * d->N * 4.0 : 0.0;

    // LDS-DMA pipelined kernel: bf16 operands, 16-byte aligned rows, every K source a multiple of 64 (the rows of
    // the contraction axis must not need zero fill); everything else runs on the generic register-staged kernel.
    const bool pipe_ok = bf && a16 && g_use_pipe && src.all_k64 && K > 0 && src.all_vec;
    const bool single = d->K2 == 0 && d->n_extra == 0;
    // fused bias gradient: only the pipelined dW form sums its A image; anything else gets explicit column sums below
    bool all_k32 = K > 0;  // (the exact-f32 pipelined kernel: f32 operands, every K source a multiple of 32)
    for (int i = 0; i < g.nsrc; ++i) all_k32 = all_k32 && (g.K[i] % 32 == 0);
    const bool pipe32_ok = !bf && !a16 && g_use_pipe && all_k32 && src.all_vec;
    const bool bias_fused = (pipe_ok || pipe32_ok) && d->dbias && d->transA && single;
    if (d->dbias) {
        EGK_REQUIRE(d->transA && single, "egk_gemm: dbias needs the dW form (transA, single K source)");
        const int64_t slab = g.splitk > 1 ? (int64_t)g.splitk * d->M * d->N * 4 : 0;
        const int64_t need = bias_fused ? slab + (g.splitk > 1 ? (int64_t)g.splitk * d->M * 4 : 0)
                                        : slab + (int64_t)egk_colsum_ws_len(d->K1, d->M) * 4;
        EGK_REQUIRE(d->ws && d->ws_bytes >= need, "egk_gemm: workspace too small for dbias (%lld bytes needed)", (long long)need);
        if (bias_fused) {
            g.dbias = d->dbias;
            g.ws_bias = g.splitk > 1 ? (float*)((char*)d->ws + slab) : nullptr;
        } else if (!query_blocks) {  // explicit column sums of dY (= A1 as stored: [K1 rows, M columns])
            const int rc = egk_colsum(stream, d->A1, d->lda1, d->K1, d->M, d->dbias, 1, (float*)((char*)d->ws + slab), d->a_dtype);
            if (rc) return rc;
        }
    }
    if (pipe_ok) {
        ensure_lds_attr();
        dim3 pblock(NTHREADS);
        // Variant policy (g_use_pipe == 1), by the number of 128 x 128 workgroups the launch would have:
        //   <= 256 : one workgroup (one wave per SIMD) per CU would walk K at a fifth of the MFMA rate (DMA issue, LDS
        //            reads and MFMAs of the lone wave serialise): two wave groups per workgroup instead (5);
        //   >  256 : 128 x 128 tiles, two 4-wave workgroups per CU (3).
        // The 256 x 128 tile (6) is kept as a measured alternative: it fetches a third fewer bytes per flop, but the
        // rate of this loop is set by LDS fragment reads + MFMA issue per 64 x 64 wave patch, not by the bytes a CU
        // takes in -- it is 3-10 % slower than (3) up to 4096^3 and 5 % faster at 8192^3 (tools/gemm_bench.py --layouts).
        const int nwg128 = g.tiles_m * g.tiles_n * g.splitk;
        const int nkt_slab = cdiv(K / 64, g.splitk);
        int variant = g_use_pipe;
        if (variant == 1) {
            variant = nwg128 > 256 ? 3 : (nkt_slab >= 4 ? 5 : 3);
            if (!d->transA && g.splitk == 1) {  // row-major A: the tile height is free (MFMA row fragments per wave)
                const long long t128 = (long long)cdiv(g.M, 128) * g.tiles_n, t96 = (long long)cdiv(g.M, 96) * g.tiles_n;
                const long long t64 = (long long)cdiv(g.M, 64) * g.tiles_n;
                if (nwg128 > 256) {
                    // a CU runs its share of the tiles two at a time at a fixed intake: the launch ends after
                    // ceil(tiles / 256) tiles' worth of bytes on the fullest CU (6144 x 1024: 2 x 32 KiB per K tile with
                    // 384 tiles of 128 rows, 2 x 28 KiB with 512 tiles of 96 rows)
                    if (((t96 + 255) / 256) * 28 < ((t128 + 255) / 256) * 32) variant = 8;
                    // ... or ONE 8-wave workgroup per CU on 192 x 128 tiles with a 3-stage ring (two K tiles = 80 KiB in flight
                    // per CU): 40 KiB per K tile and round of 256 tiles -- 6144 x 1024 is exactly one round (256 tiles) where the
                    // 96-row tiles are two co-resident workgroups of 28 KiB each
                    const long long t192 = (long long)cdiv(g.M, 192) * g.tiles_n;
                    const long long cur = variant == 8 ? ((t96 + 255) / 256) * 28 : ((t128 + 255) / 256) * 32;
                    // (ONE round only: 8192 x 1024 -- 344 tiles, two rounds of which the second is a third full -- measured 32.6 us
                    //  against 22.3 on 128-row tiles; BASELINE config 5's 16384 rows 3.08 against 2.92 ms per step)
                    if (g_r192 && t192 > 192 && t192 <= 256 && 40 < cur) variant = 16;
                    // (several rounds of this tile with its loader waves -- 16384 x 1024: 688 tiles, 2.7 rounds -- measured 2.938 -> 2.901 ms
                    //  for BASELINE config 5 on one box and 2.93 -> 3.03 on two others: one round only)
                } else if (t64 <= 256 && t64 > t128) {
                    // at most 128 tiles: 64-row tiles put one 4-wave workgroup on twice as many CUs instead of one
                    // 8-wave (two wave groups) workgroup on half of them (2048 x 1024 x 1024: 10.6 vs 12.4 us)
                    variant = 11;
                    // ... and BOTH where the epilogue needs no finished tile in one wave group (no LayerNorm statistics, no row
                    // gather): 64-row tiles on every CU, two wave groups per workgroup walking alternate K tiles -- the lone
                    // 4-wave workgroup of (11) leaves its CU's DMA / LDS / matrix phases unoverlapped
                    if (g_wg2_rows64 && !d->st_mode && nkt_slab >= 8) variant = 12;
                }
            }
            // large outputs with a long K walk: 256 x 256 tiles, one 8-wave workgroup per CU, when whole tiles fill at least
            // 90 % of the CU slots of the rounds they take.  Measured against the 128 x 128 kernel (tools/gemm_big.py):
            // 16384 x 1024 x 4608: 126 vs 145 us; 4096^3: 113 vs 128; 8192^3: 836 vs 1178; but 16384 x 1024 x 2048: 76 vs 74
            // and x 1024: 45 vs 45 (16 K tiles: prologue and epilogue of the big tile weigh as much as its loop saves),
            // 24576 x 1024 (384 tiles, 1.5 rounds): 78 vs 65.  6144 x 1024 (96 tiles) never qualifies.
            const long long t256 = (long long)cdiv(g.M, 256) * cdiv(g.N, 256);
            if (g.splitk == 1 && !g.dbias && g.M % 256 == 0 && g.N % 256 == 0 && t256 >= 192 && K >= 3072 &&
                10 * t256 >= 9 * 256 * ((t256 + 255) / 256))
                variant = 7;
            // ... and 192 x 256 tiles (row-major A) where THEY fill their rounds and the 256-row tiles do not: 6144 x 4096 is 384
            // tiles of 256 rows (1.5 rounds: the makespan of two) but 512 of 192 rows (two whole rounds of 3/4 the height)
            const long long t192 = (long long)cdiv(g.M, 192) * cdiv(g.N, 256);
            // (tools/round5/pp_bench.py, us: 6144 x 4096 x 4608 199 vs 211-263 on 128-row tiles, x 4096 180 vs 190, the dX form 185 vs
            //  190, 6144 x 4096 x 1024 63.3 vs 66.7; 6144 x 1024 outputs are 128 such tiles and stay on 96-row tiles)
            if (variant != 7 && !d->transA && g.splitk == 1 && !g.dbias && g.M % 192 == 0 && g.N % 256 == 0 && t192 >= 192 && K >= 1024 &&
                10 * t192 >= 9 * 256 * ((t192 + 255) / 256))
                variant = 15;
        } else if ((variant == 8 || variant == 11 || variant == 12 || variant == 16) &&
                   (d->transA || (variant == 16 && g.splitk > 1))) {
            variant = 3;  // the forced variants exist for row-major A only (16: unsplit)
        } else if (variant == 7 && (g.dbias || g.M % 256 != 0 || g.N % 256 != 0)) {
            variant = 3;  // whole 256 x 256 tiles only, no fused bias gradient
        } else if (variant == 15 && (d->transA || g.dbias || g.M % 192 != 0 || g.N % 256 != 0)) {
            variant = 3;  // whole 192 x 256 tiles of a row-major A only
        }
        if (variant != 3 && variant != 5 && variant != 7 && variant != 8 && variant != 11 && variant != 12 && variant != 15 && variant != 16)
            variant = 3;  // (a forced value that names no tile variant: the 2-stage 128 x 128 kernel)
        if (variant == 12 && d->st_mode) variant = 11;  // (forced by the knob: the epilogue statistics win)
        g.tiles_m = variant == 8 ? cdiv(g.M, 96) : (variant == 11 || variant == 12) ? cdiv(g.M, 64) : variant == 7 ? cdiv(g.M, 256)
                    : (variant == 15 || variant == 16) ? cdiv(g.M, 192) : cdiv(g.M, BM);
        if (variant == 7 || variant == 15) g.tiles_n = cdiv(g.N, 256);
#ifdef EGK_GEMM_STAMPS
        if (variant != 7 && !g.dbias && d->ws) {
            const int64_t slab = g.splitk > 1 ? (int64_t)g.splitk * d->M * d->N * 4 : 0;
            if (d->ws_bytes >= slab + (int64_t)g.tiles_m * g.tiles_n * g.splitk * 64) g.ws_bias = (float*)((char*)d->ws + slab);
        }
        if (variant == 7 && g.splitk == 1 && !g.dbias && d->ws && d->ws_bytes >= (int64_t)g.tiles_m * g.tiles_n * 8 * 64)
            g.ws_bias = (float*)d->ws;  // per-wave cycle stamps land in the caller's workspace
#endif
        {  // near-square XCD patches: group_m ~ sqrt(workgroups per XCD), inside one slab
            const int tiles = g.tiles_m * g.tiles_n;
            int per_xcd = cdiv(tiles * g.splitk, 8);
            if (per_xcd > tiles) per_xcd = tiles;
            int gm = 1;
            while ((gm + 1) * (gm + 1) <= per_xcd) ++gm;
            g.group_m = gm < g.tiles_m ? gm : g.tiles_m;
            // Row-major A (forward / dX forms: the output rows ARE the rows of the chain's activations): XCD x takes the tile rows
            // [x * tiles_m / 8, (x + 1) * tiles_m / 8) with ALL their tile columns -- the contiguous eighth of the rows that the row
            // kernels in front of and behind this launch give to XCD x (common.h, row_walk): its A strips were written, and its output
            // rows will be read, by workgroups of the SAME XCD, i.e. through that XCD's own L2 instead of across the fabric
            // (tools/exp/xcd_affinity.hip: 4.0-4.4 us against 8.3-11.2 us for a 12.6 MB hand-off).  Every XCD then streams the whole
            // B operand (weights: 2-9 MB, shared by its 32 workgroups K tile by K tile).
            if (g_row_affinity && !d->transA && g.splitk == 1 && g.tiles_m % 8 == 0) g.group_m = g.tiles_m / 8;
            if (g_group_m_override > 0) g.group_m = g_group_m_override < g.tiles_m ? g_group_m_override : g.tiles_m;
        }

        // segment statistics in the epilogue: the 4-wave variants that write their tile out through LDS in whole rows, unsplit,
        // and tiles that span at most two row segments
        const int tile_rows = variant == 8 ? 96 : (variant == 11 || variant == 12) ? 64 : variant == 16 ? 192 : 128;
        const bool st_ok = (variant == 3 || variant == 8 || variant == 11 || variant == 16) && g.splitk == 1 &&
                           epilogue_rows_ok(g) && d->st_min_seg_rows >= tile_rows &&
                           (d->st_mode != 2 || (aligned16(d->st_x) && d->st_ldx % (g.c_bf16 ? 8 : 4) == 0 && aligned16(d->st_w) && aligned16(d->st_b)));
        if (query_blocks) {
            *query_blocks = (d->st_mode && st_ok) ? g.tiles_m * g.tiles_n : 0;
            return 0;
        }
        if (d->st_mode && !st_ok) {
            set_error("egk_gemm: st_mode %d is not available for this launch (ask egk_gemm_stats_blocks first)", d->st_mode);
            return EGK_EUNSUPPORTED;
        }
        dim3 pgrid(g.tiles_m * g.tiles_n * g.splitk);
#define EGK_PIPE(TA, TB)                                                                                                  \
    do {                                                                                                                  \
        if (variant == 16 && g_r192_loaders && nkt_slab >= 4) {                                                           \
            hipLaunchKernelGGL((gemm_pipe_kernel<3, false, TB, 1, 2, 3, false, 4>), pgrid, dim3(2 * NTHREADS + 256), 3 * 40960, s, g); \
        } else if (variant == 16) {                                                                                       \
            hipLaunchKernelGGL((gemm_pipe_kernel<3, false, TB, 1, 2, 3>), pgrid, dim3(2 * NTHREADS), 3 * 40960, s, g);    \
        } else if (variant == 15) {                                                                                       \
            hipLaunchKernelGGL((gemm_big_kernel<false, TB, 8, 6>), pgrid, dim3(512), 131072, s, g);                       \
        } else if (variant == 7) {                                                                                        \
            hipLaunchKernelGGL((gemm_big_kernel<TA, TB>), pgrid, dim3(512), 131072, s, g);                                \
        } else if (variant == 12) {                                                                                       \
            hipLaunchKernelGGL((gemm_pipe_kernel<2, false, TB, 2, 1, 2>), pgrid, dim3(2 * NTHREADS), 4 * 24576, s, g);    \
        } else if (variant == 11) {                                                                                       \
            hipLaunchKernelGGL((gemm_pipe_kernel<2, false, TB, 1, 1, 2>), pgrid, pblock, 2 * 24576, s, g);                \
        } else if (variant == 8) {                                                                                        \
            hipLaunchKernelGGL((gemm_pipe_kernel<2, false, TB, 1, 1, 3>), pgrid, pblock, 2 * 28672, s, g);                \
        } else if (variant == 5)                                                                                          \
            hipLaunchKernelGGL((gemm_pipe_kernel<2, TA, TB, 2, 1>), pgrid, dim3(2 * NTHREADS), 4 * 32768, s, g);          \
        else hipLaunchKernelGGL((gemm_pipe_kernel<2, TA, TB, 1, 1>), pgrid, pblock, 2 * 32768, s, g);                     \
    } while (0)
        {
            ProfScope prof((variant == 7 || variant == 15) ? KID_GEMM_BF16_NN_T256 + layout : variant == 8 ? KID_GEMM_BF16_NN_R96 + layout : variant == 11 ? KID_GEMM_BF16_NN_R64 + layout
                                       : variant == 16 ? KID_GEMM_BF16_NN_R192 + layout
                                       : ((variant == 5 || variant == 12) ? KID_GEMM_BF16_NN_G2 : KID_GEMM_BF16_NN) + layout, s, flops, bytes);
            if (!d->transA && !d->transB) EGK_PIPE(false, false);
            else if (!d->transA && d->transB) EGK_PIPE(false, true);
            else if (d->transA && d->transB) EGK_PIPE(true, true);
            else EGK_PIPE(true, false);
        }
#undef EGK_PIPE
        const bool defer = take_defer_reduce();
        if (defer)
            EGK_REQUIRE(g.splitk > 1 && !g.accumulate && g.act == 0 && !g.residual && g.alpha == 1.f && !g.c_bf16 && !g.dbias,
                        "egk_gemm_defer_reduce_next: a split launch with a plain f32 result (bias only)");
        if (g.splitk > 1 && !defer) {
            ProfScope prof(KID_GEMM_SPLITK_REDUCE, s, 0, slab_bytes);
            const long long total = (long long)g.M * g.N;
            const long long work = (total + 3) / 4;  // element groups of 4 (the vector path; the scalar path strides)
            hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048)),
                               dim3(256), 0, s, g);
        }
        return check_launch("egk_gemm");
    }
    // exact-f32 pipelined kernel: f32 operands, 16-byte aligned rows, every K source a multiple of 32
    if (pipe32_ok) {
        ensure_lds_attr();
        // 96-row tiles (row-major A) when they load the CUs more evenly: the matrix pipes of a CU are shared by its co-resident
        // workgroups, so a launch lasts as long as the ROWS on its fullest CU
        bool r96 = false;
        if (!d->transA && g_use_pipe != 3) {
            const long long t128 = (long long)cdiv(g.M, 128) * g.tiles_n * g.splitk, t96 = (long long)cdiv(g.M, 96) * g.tiles_n * g.splitk;
            r96 = g_use_pipe == 8 || ((t96 + 255) / 256) * 96 < ((t128 + 255) / 256) * 128;
        }
        if (r96) g.tiles_m = cdiv(g.M, 96);
        {  // near-square XCD patches, as for the bf16 kernel
            const int tiles = g.tiles_m * g.tiles_n;
            int per_xcd = cdiv(tiles * g.splitk, 8);
            if (per_xcd > tiles) per_xcd = tiles;
            int gm = 1;
            while ((gm + 1) * (gm + 1) <= per_xcd) ++gm;
            g.group_m = gm < g.tiles_m ? gm : g.tiles_m;
            if (g_row_affinity && !d->transA && g.splitk == 1 && g.tiles_m % 8 == 0) g.group_m = g.tiles_m / 8;  // (as for the bf16 kernel)
        }
        const bool st_ok = g.splitk == 1 && epilogue_rows_ok(g) && d->st_min_seg_rows >= (r96 ? 96 : 128) &&
                           (d->st_mode != 2 || (aligned16(d->st_x) && d->st_ldx % 4 == 0 && aligned16(d->st_w) && aligned16(d->st_b)));
        if (query_blocks) {
            *query_blocks = (d->st_mode && st_ok) ? g.tiles_m * g.tiles_n : 0;
            return 0;
        }
        if (d->st_mode && !st_ok) {
            set_error("egk_gemm: st_mode %d is not available for this launch (ask egk_gemm_stats_blocks first)", d->st_mode);
            return EGK_EUNSUPPORTED;
        }
        const dim3 pgrid(g.tiles_m * g.tiles_n * g.splitk), pblock(NTHREADS);
        {
            ProfScope prof(KID_GEMM_F32_NN + layout, s, flops, bytes);
            if (r96 && !d->transB) hipLaunchKernelGGL((gemm_pipe_f32_kernel<false, false, 3>), pgrid, pblock, 2 * 28672, s, g);
            else if (r96) hipLaunchKernelGGL((gemm_pipe_f32_kernel<false, true, 3>), pgrid, pblock, 2 * 28672, s, g);
            else if (!d->transA && !d->transB) hipLaunchKernelGGL((gemm_pipe_f32_kernel<false, false>), pgrid, pblock, 65536, s, g);
            else if (!d->transA && d->transB) hipLaunchKernelGGL((gemm_pipe_f32_kernel<false, true>), pgrid, pblock, 65536, s, g);
            else if (d->transA && d->transB) hipLaunchKernelGGL((gemm_pipe_f32_kernel<true, true>), pgrid, pblock, 65536, s, g);
            else hipLaunchKernelGGL((gemm_pipe_f32_kernel<true, false>), pgrid, pblock, 65536, s, g);
        }
        EGK_REQUIRE(!take_defer_reduce(), "egk_gemm_defer_reduce_next: only the bf16-operand pipelined launches leave their slabs");
        if (g.splitk > 1) {
            ProfScope prof(KID_GEMM_SPLITK_REDUCE, s, 0, slab_bytes);
            const long long total = (long long)g.M * g.N;
            const long long work = (total + 3) / 4;
            hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048)),
                               dim3(256), 0, s, g);
        }
        return check_launch("egk_gemm");
    }
    if (query_blocks) return 0;  // (the generic kernel has no statistics epilogue: 0 blocks, not ok)
    if (d->st_mode) {
        set_error("egk_gemm: st_mode %d is not available on the generic kernel (ask egk_gemm_stats_blocks first)", d->st_mode);
        return EGK_EUNSUPPORTED;
    }
    dim3 grid(g.tiles_m * g.tiles_n, g.splitk);
    {
        ProfScope prof(bf ? KID_GEMM_BF16_GENERIC : KID_GEMM_F32_NN + layout, s, flops, bytes);
        if (!bf) launch_layout<false, float, float>(d, grid, s, g);
        else if (a16) launch_layout<true, bf16_t, bf16_t>(d, grid, s, g);
        else launch_layout<true, float, float>(d, grid, s, g);
    }
    EGK_REQUIRE(!take_defer_reduce(), "egk_gemm_defer_reduce_next: only the bf16-operand pipelined launches leave their slabs");
    if (g.splitk > 1) {
        ProfScope prof(KID_GEMM_SPLITK_REDUCE, s, 0, slab_bytes);
        const long long total = (long long)g.M * g.N;
        hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)),
                           dim3(256), 0, s, g);
    }
    return check_launch("egk_gemm");
}

// The reduce launch of a split contraction by itself: C[m, n] = sum_z ws[z][m][n] (slab order) + bias[n] -- for slabs left behind
// by egk_gemm_defer_reduce_next that no slab-aware row kernel consumed.
extern "C" int egk_gemm_reduce_slabs(egk_stream_t stream, const float* ws, int32_t splitk, int32_t M, int32_t N, const float* bias,
                                     float* C, int64_t ldc) {
    EGK_REQUIRE(ws && C && splitk >= 2 && M >= 0 && N >= 0 && ldc >= N, "egk_gemm_reduce_slabs: bad arguments");
    if (M == 0 || N == 0) return 0;
    GemmArgs g{};  // (value-initialised: every epilogue feature off)
    g.M = M; g.N = N;
    g.C = C; g.ldc = ldc;
    g.c_vec = aligned16(C) && ldc % 4 == 0;
    g.alpha = 1.f;
    g.bias = bias;
    g.splitk = splitk;
    g.ws = const_cast<float*>(ws);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_GEMM_SPLITK_REDUCE, s, 0, 4.0 * (splitk + 1) * M * N);
    const long long work = ((long long)M * N + 3) / 4;
    hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048)), dim3(256), 0, s, g);
    return check_launch("egk_gemm_reduce_slabs");
}

// ---- grouped contractions ---------------------------------------------------------------------------------------------
// Fill the device-side argument block of one problem of a grouped launch (bf16 operands on the pipelined kernel only).
static int fill_group_args(const egk_gemm_desc* d, GemmArgs& g) {
    EGK_REQUIRE(d->M > 0 && d->N > 0 && d->K1 > 0 && d->K2 >= 0, "egk_gemm_grouped: empty problem");
    const bool f32p = d->a_dtype == EGK_F32 && d->b_dtype == EGK_F32 && d->compute == EGK_COMPUTE_F32;
    EGK_REQUIRE(f32p || (d->a_dtype == EGK_BF16 && d->b_dtype == EGK_BF16 && d->compute == EGK_COMPUTE_BF16),
                "egk_gemm_grouped: bf16 operands on the bf16 MFMA path, or f32 operands on the exact-f32 path");
    EGK_REQUIRE(d->splitk <= 1, "egk_gemm_grouped: no split-K in a grouped launch");
    EGK_REQUIRE(!(d->accumulate && d->c_dtype != EGK_F32), "egk_gemm_grouped: accumulate needs an f32 C");
    EGK_REQUIRE(d->C, "egk_gemm_grouped: null pointer");
    EGK_REQUIRE(!d->dbias || (d->transA && d->K2 == 0 && d->n_extra == 0), "egk_gemm_grouped: dbias needs the dW form (transA, single K source)");
    g.M = d->M; g.N = d->N;
    SourceInfo src;
    {
        const int rc = fill_sources(d, g, f32p ? 4 : 8, f32p ? 4 : 8, src, "egk_gemm_grouped");
        if (rc) return rc;
    }
    if (f32p) {
        for (int i = 0; i < g.nsrc; ++i) EGK_REQUIRE(g.K[i] % 32 == 0, "egk_gemm_grouped: f32 K sources must be multiples of 32");
    } else {
        EGK_REQUIRE(src.all_k64, "egk_gemm_grouped: K sources must be multiples of 64");
    }
    EGK_REQUIRE(src.all_vec, "egk_gemm_grouped: operand rows must be 16-byte aligned");
    g.C = d->C; g.ldc = d->ldc;
    g.c_bf16 = d->c_dtype == EGK_BF16;
    g.c_vec = g.c_bf16 ? ((reinterpret_cast<uintptr_t>(g.C) & 7) == 0 && g.ldc % 4 == 0) : (aligned16(g.C) && g.ldc % 4 == 0);
    g.accumulate = d->accumulate; g.act = d->act; g.alpha = d->alpha;
    g.bias = d->bias; g.residual = d->residual; g.ldr = d->ldr;
    g.r_bf16 = d->r_dtype == EGK_BF16;
    g.r_vec = g.residual && (g.r_bf16 ? ((reinterpret_cast<uintptr_t>(g.residual) & 7) == 0 && g.ldr % 4 == 0)
                                      : (aligned16(g.residual) && g.ldr % 4 == 0));
    g.splitk = 1;
    g.ws = nullptr;
    g.dbias = d->dbias; g.ws_bias = nullptr;
    g.rows_epilogue = g_rows_epilogue;
    g.st_mode = 0; g.st_nseg = 0; g.st_seg_ptr = nullptr; g.st_ws = nullptr; g.st_x = nullptr; g.st_ldx = 0;
    g.st_stats = nullptr; g.st_w = nullptr; g.st_b = nullptr; g.st_slope = 0.f;
    EGK_REQUIRE(d->st_mode == 0, "egk_gemm_grouped: no segment statistics in a grouped launch");
    return 0;
}

extern "C" int egk_gemm_grouped(egk_stream_t stream, const egk_gemm_desc* descs, int32_t count) {
    EGK_REQUIRE(descs != nullptr && count >= 1 && count <= MAX_GROUPS, "egk_gemm_grouped: 1 .. %d problems", MAX_GROUPS);
    if (count == 1) return egk_gemm(stream, descs);
    hipStream_t s = (hipStream_t)stream;
    GemmGroup gg;
    const bool ta = descs[0].transA != 0, tb = descs[0].transB != 0;
    const bool f32g = descs[0].compute == EGK_COMPUTE_F32;
    const bool f16 = descs[0].op_f16 != 0;
    EGK_REQUIRE(!f16 || (!ta && !tb && !f32g), "egk_gemm_grouped: op_f16 takes row-major 16-bit operands");
    double flops = 0, bytes = 0;
    long long t128 = 0, t96 = 0, t64 = 0, t256 = 0, t192 = 0;
    int min_nkt = 1 << 30, any_extra = 0;
    for (int i = 0; i < count; ++i) {
        const egk_gemm_desc* d = descs + i;
        EGK_REQUIRE((d->transA != 0) == ta && (d->transB != 0) == tb, "egk_gemm_grouped: the problems must share one layout");
        EGK_REQUIRE((d->compute == EGK_COMPUTE_F32) == f32g, "egk_gemm_grouped: the problems must share one compute type");
        EGK_REQUIRE((d->op_f16 != 0) == f16, "egk_gemm_grouped: the problems must agree on op_f16");
        EGK_REQUIRE(!f16 || (!d->st_mode && !d->dbias && d->n_extra == 0), "egk_gemm_grouped: op_f16 has no statistics / bias-gradient epilogue");
        const int rc = fill_group_args(d, gg.p[i]);
        if (rc) return rc;
        int K = d->K1 + d->K2;
        for (int x = 0; x < d->n_extra; ++x) K += d->xK[x];
        flops += 2.0 * d->M * d->N * K;
        bytes += 2.0 * ((double)d->M * K + (double)d->N * K) + (gg.p[i].c_bf16 ? 2.0 : 4.0) * d->M * d->N;
        const int tn = cdiv(d->N, BN);
        t128 += (long long)cdiv(d->M, 128) * tn; t96 += (long long)cdiv(d->M, 96) * tn; t64 += (long long)cdiv(d->M, 64) * tn;
        t256 += (long long)cdiv(d->M, 256) * tn;
        t192 += (long long)cdiv(d->M, 192) * tn;
        any_extra |= d->n_extra > 0;
        min_nkt = K / 64 < min_nkt ? K / 64 : min_nkt;
    }
    ensure_lds_attr();
    if (f32g) {  // exact-f32 problems: 128 x 128 tiles, XCD-packed placement
        int total = 0;
        for (int i = 0; i < count; ++i) {
            GemmArgs& g = gg.p[i];
            g.tiles_m = cdiv(g.M, BM);
            g.tiles_n = cdiv(g.N, BN);
            total += g.tiles_m * g.tiles_n;
        }
        for (int i = 0; i < count; ++i) {
            GemmArgs& g = gg.p[i];
            const int tiles = g.tiles_m * g.tiles_n;
            int per_xcd = cdiv(total, 8), gm = 1;
            if (per_xcd > tiles) per_xcd = tiles;
            while ((gm + 1) * (gm + 1) <= per_xcd) ++gm;
            g.group_m = gm < g.tiles_m ? gm : g.tiles_m;
        }
        for (int i = count; i < MAX_GROUPS; ++i) gg.p[i] = gg.p[0];
        gg.packed = 1; gg.count = count; gg.total = total;
        const dim3 pgrid((total + 7) / 8 * 8), pblock(NTHREADS);
        const int layout = ta ? (tb ? 2 : 3) : (tb ? 1 : 0);
        EGK_REQUIRE(!(ta && !tb), "egk_gemm_grouped: the tn layout is not instantiated");
        ProfScope prof(KID_GEMM_F32_NN + layout, s, flops, 2.0 * bytes);  // (bytes above count 2 per operand element)
        if (!ta && !tb) hipLaunchKernelGGL((gemm_pipe_f32_group_kernel<false, false>), pgrid, pblock, 65536, s, gg);
        else if (!ta) hipLaunchKernelGGL((gemm_pipe_f32_group_kernel<false, true>), pgrid, pblock, 65536, s, gg);
        else hipLaunchKernelGGL((gemm_pipe_f32_group_kernel<true, true>), pgrid, pblock, 65536, s, gg);
        return check_launch("egk_gemm_grouped");
    }
    // one tile variant for the whole launch, by the policy of egk_gemm applied to the TOTAL tile count
    int variant = t128 > 256 ? 3 : (min_nkt >= 4 ? 5 : 3);
    if (!ta) {
        if (t128 > 256) {
            if (((t96 + 255) / 256) * 28 < ((t128 + 255) / 256) * 32) variant = 8;
        } else if (t64 <= 256 && t64 > t128) {
            variant = 11;
        }
    }
    // weight-gradient groups (A^T B): a CU works on one 128 x 128 workgroup at a time (272 registers per lane), so a launch of 257 ..
    // 512 tiles takes two rounds whatever its tile count; when that count leaves many CUs idle in the second round (the pooling's
    // three weight gradients: 416 tiles) and the 256 x 128 tiling fits one workgroup per CU, the tall tile (8 waves, 3-stage ring,
    // 48 KiB per K tile for twice the flops) is one round instead.  Long K walks only: same box, alternating, 6144 rows 1.291 -> 1.284 ms, 16384 rows
    // 2.947 -> 2.934, 2048 rows 0.842 -> 0.863 (its 3-stage fill and 256-row epilogue outweigh 32 K tiles).
    if (ta && tb && g_tt_tall && t128 > 256 && t128 <= 448 && t256 <= 256 && min_nkt >= 64) variant = 13;
    // row-major A: the 192 x 128 tile with loader waves (egk_gemm's variant 16) when the group is ONE round of such tiles that more
    // than half fills the chip and 128- / 96-row tiles would take more than one (the projection heads of three task batches:
    // 64 + 2048 + 2048 rows = 184 tiles against 264 / 360)
    // (not for the three-product groups of the precise pass: BASELINE config 4 2.113 -> 2.124 ms with them)
    if (!ta && !f16 && !any_extra && g_r192 && g_r192_loaders && g_group_r192 && t192 > 128 && t192 <= 256 && t128 > 256 && min_nkt >= 8) variant = 16;
    if (g_use_pipe == 3 || g_use_pipe == 5) variant = g_use_pipe;
    if (g_use_pipe == 13 && ta && tb) variant = 13;
    if ((g_use_pipe == 8 || g_use_pipe == 11) && !ta) variant = g_use_pipe;
    if (f16) variant = 3;  // (one instantiation: 128 x 128 tiles, one wave group)
    int max_wg = 0, total = 0;
    for (int i = 0; i < count; ++i) {
        GemmArgs& g = gg.p[i];
        g.tiles_m = variant == 8 ? cdiv(g.M, 96) : variant == 11 ? cdiv(g.M, 64) : variant == 13 ? cdiv(g.M, 256) : variant == 16 ? cdiv(g.M, 192) : cdiv(g.M, BM);
        g.tiles_n = cdiv(g.N, BN);
        total += g.tiles_m * g.tiles_n;
    }
    const bool packed = g_group_packed != 0;
    for (int i = 0; i < count; ++i) {
        GemmArgs& g = gg.p[i];
        const int tiles = g.tiles_m * g.tiles_n;
        // near-square patches of what ONE XCD works on: its share of this problem (spread placement) or of the whole launch
        int per_xcd = packed ? cdiv(total, 8) : cdiv(tiles, 8), gm = 1;
        if (per_xcd > tiles) per_xcd = tiles;
        while ((gm + 1) * (gm + 1) <= per_xcd) ++gm;
        g.group_m = gm < g.tiles_m ? gm : g.tiles_m;
        max_wg = tiles > max_wg ? tiles : max_wg;
    }
    for (int i = count; i < MAX_GROUPS; ++i) gg.p[i] = gg.p[0];
    gg.packed = packed ? 1 : 0; gg.count = count; gg.total = total;
    const dim3 pgrid = packed ? dim3((total + 7) / 8 * 8) : dim3((max_wg + 7) / 8 * 8, count);
    const dim3 pblock(NTHREADS);
    const int layout = ta ? (tb ? 2 : 3) : (tb ? 1 : 0);
    EGK_REQUIRE(!(ta && !tb), "egk_gemm_grouped: the tn layout is not instantiated");
#define EGK_PIPE_G(TA, TB)                                                                                                  \
    do {                                                                                                                    \
        if (variant == 5) hipLaunchKernelGGL((gemm_pipe_group_kernel<2, TA, TB, 2, 1>), pgrid, dim3(2 * NTHREADS), 4 * 32768, s, gg); \
        else hipLaunchKernelGGL((gemm_pipe_group_kernel<2, TA, TB, 1, 1>), pgrid, pblock, 2 * 32768 + pad, s, gg);          \
    } while (0)
    // (experiment) a weight-gradient group that runs BESIDE the backward chain may be held to one workgroup per CU by asking for
    // more LDS than half a CU has: the chain's launches then always find registers and LDS on every CU
    int pad = 0;
    if (ta && tb && g_group_tt_pad_kb > 0) {
        bool side = true;
        for (int i = 0; i < count; ++i) side = side && gg.p[i].N <= 1024;
        if (side) pad = g_group_tt_pad_kb * 1024;
    }
    {
        ProfScope prof(KID_GEMM_BF16_GROUP_NN + layout, s, flops, bytes);  // (layout 0 nn, 1 nt, 2 tt; the tall weight-gradient tile included)
        if (f16) {
            ensure_lds_attr();
            hipLaunchKernelGGL((gemm_pipe_group_kernel<2, false, false, 1, 1, 4, true>), pgrid, pblock, 2 * 32768, s, gg);
        } else if (!ta && variant == 16) {
            if (!tb) hipLaunchKernelGGL((gemm_pipe_group_kernel<3, false, false, 1, 2, 3, false, 4>), pgrid, dim3(2 * NTHREADS + 256), 3 * 40960, s, gg);
            else hipLaunchKernelGGL((gemm_pipe_group_kernel<3, false, true, 1, 2, 3, false, 4>), pgrid, dim3(2 * NTHREADS + 256), 3 * 40960, s, gg);
        } else if (!ta && variant == 11) {
            if (!tb) hipLaunchKernelGGL((gemm_pipe_group_kernel<2, false, false, 1, 1, 2>), pgrid, pblock, 2 * 24576, s, gg);
            else hipLaunchKernelGGL((gemm_pipe_group_kernel<2, false, true, 1, 1, 2>), pgrid, pblock, 2 * 24576, s, gg);
        } else if (!ta && variant == 8) {
            if (!tb) hipLaunchKernelGGL((gemm_pipe_group_kernel<2, false, false, 1, 1, 3>), pgrid, pblock, 2 * 28672, s, gg);
            else hipLaunchKernelGGL((gemm_pipe_group_kernel<2, false, true, 1, 1, 3>), pgrid, pblock, 2 * 28672, s, gg);
        } else if (!ta && !tb) EGK_PIPE_G(false, false);
        else if (!ta && tb) EGK_PIPE_G(false, true);
        else if (variant == 13) hipLaunchKernelGGL((gemm_pipe_group_kernel<3, true, true, 1, 2>), pgrid, dim3(2 * NTHREADS), 3 * 49152, s, gg);
        else EGK_PIPE_G(true, true);
    }
#undef EGK_PIPE_G
    return check_launch("egk_gemm_grouped");
}
