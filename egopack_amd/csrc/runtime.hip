// Library runtime: error strings and the built-in per-kernel HIP-event profiler.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace egk {

// one-shot split tee (egk_tee_split_next): per host thread, consumed by the next row-kernel launcher
static thread_local SplitTee g_tee = {nullptr, nullptr, 0};
void arm_split_tee(bf16_t* hi, bf16_t* lo, long long ld) { g_tee = SplitTee{hi, lo, ld}; }
SplitTee take_split_tee() {
    const SplitTee t = g_tee;
    g_tee = SplitTee{nullptr, nullptr, 0};
    return t;
}
bool split_tee_armed() { return g_tee.lo != nullptr; }
static thread_local SlabInput g_slab = {nullptr, nullptr, nullptr};
static thread_local bool g_defer_reduce = false;
void arm_slab_input(const float* x2, const float* bias, float* x_out) { g_slab = SlabInput{x2, bias, x_out}; }
SlabInput take_slab_input() {
    const SlabInput t = g_slab;
    g_slab = SlabInput{nullptr, nullptr, nullptr};
    return t;
}
bool slab_input_armed() { return g_slab.x2 != nullptr; }
void arm_defer_reduce(bool on) { g_defer_reduce = on; }
bool take_defer_reduce() {
    const bool t = g_defer_reduce;
    g_defer_reduce = false;
    return t;
}

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static const char* const kNames[KID_COUNT] = {
    "gemm_bf16_nn", "gemm_bf16_nt", "gemm_bf16_tt", "gemm_bf16_tn",
    "gemm_f32_nn", "gemm_f32_nt", "gemm_f32_tt", "gemm_f32_tn",
    "gemm_bf16_nn_g2", "gemm_bf16_nt_g2", "gemm_bf16_tt_g2", "gemm_bf16_tn_g2", "gemm_bf16_nn_r96", "gemm_bf16_nt_r96", "gemm_bf16_nn_r64", "gemm_bf16_nt_r64",
    "gemm_bf16_nn_r192", "gemm_bf16_nt_r192",
    "gemm_bf16_nn_t256", "gemm_bf16_nt_t256", "gemm_bf16_tt_t256", "gemm_bf16_tn_t256",
    "gemm_bf16_group_nn", "gemm_bf16_group_nt", "gemm_bf16_group_tt",
    "gemm_bf16_generic", "gemm_splitk_reduce",
    "colsum", "rowln_fwd", "rowln_bwd", "rowln_bwd_reduce",
    "graphln_stats", "graphln_fwd", "graphln_bwd_stats", "graphln_bwd", "graphln_bwd_reduce",
    "pe_add", "csr_gather", "gather_max_fwd", "gather_max_bwd", "segmax_fwd", "segmax_bwd",
    "row_inv_norm", "topk", "scatter_add_f64", "ce_fwd", "ce_bwd", "bce_fwd", "bce_bwd",
    "dropout_fwd", "dropout_bwd", "relu_gate", "cast", "axpby", "sum_scale", "adam"};

struct Pending {
    int kid;
    hipEvent_t a, b;
};
struct Totals {
    int64_t launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};

static std::mutex g_mu;
static bool g_on = false;
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_free;
static Totals g_tot[KID_COUNT];
static thread_local Pending g_cur;

bool prof_on() { return g_on; }

static hipEvent_t get_event() {
    if (!g_free.empty()) {
        hipEvent_t e = g_free.back();
        g_free.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(int kid, hipStream_t s, double flops, double bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_cur.kid = kid;
    g_cur.a = get_event();
    g_cur.b = get_event();
    g_tot[kid].launches += 1;
    g_tot[kid].flops += flops;
    g_tot[kid].bytes += bytes;
    (void)hipEventRecord(g_cur.a, s);
}

void prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_cur.b, s);
    g_pending.push_back(g_cur);
}

static void drain() {
    for (auto& p : g_pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) g_tot[p.kid].ms += ms;
        g_free.push_back(p.a);
        g_free.push_back(p.b);
    }
    g_pending.clear();
}

// one lane stamps the constant-rate wall clock (s_memrealtime, 100 MHz) into slot idx: phase boundaries of a captured
// step can be timed in place, at ~2 us per stamp, without a profiler serialising the queues
__global__ void stamp_kernel(unsigned long long* __restrict__ buf, int idx) {
    if (threadIdx.x == 0) buf[idx] = wall_clock64();
}

}  // namespace egk

using namespace egk;

extern "C" {

int egk_stamp(egk_stream_t stream, uint64_t* buf, int32_t idx) {
    EGK_REQUIRE(buf && idx >= 0, "egk_stamp: bad arguments");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)buf, idx);
    return check_launch("egk_stamp");
}

/* Host-side helpers of the batch builders (no device work).  The reference draws its segment-sampling offsets window by
 * window with numpy's legacy ``RandomState.randint(high, size=n)`` (data/base_dataset.py:128-139).  Reproducing that stream
 * exactly lets a whole batch of windows be sampled in one call: the generator is the published MT19937 (state = 624 words +
 * position, as ``RandomState.get_state()`` hands it out; a draw regenerates the block at position 624 and tempers the next
 * word), and a bounded value is ``do v = next_uint32 & mask; while (v > high - 1)`` with mask = the smallest 2^k - 1 >= high - 1
 * (numpy/random/src/distributions: buffered_bounded_masked_uint32; a bound of 1 consumes nothing).  The callers write the
 * advanced state back with ``set_state``, so the stream continues where the per-window calls would have left it. */
namespace {
struct MT {
    uint32_t* key;
    int pos;
    inline void regenerate() {
        const uint32_t UP = 0x80000000u, LOW = 0x7fffffffu, A = 0x9908b0dfu;
        int k = 0;
        for (; k < 624 - 397; ++k) {
            const uint32_t y = (key[k] & UP) | (key[k + 1] & LOW);
            key[k] = key[k + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        for (; k < 623; ++k) {
            const uint32_t y = (key[k] & UP) | (key[k + 1] & LOW);
            key[k] = key[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        const uint32_t y = (key[623] & UP) | (key[0] & LOW);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    inline uint32_t next() {
        if (pos >= 624) regenerate();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    inline uint32_t bounded(uint32_t rng, uint32_t mask) {
        uint32_t v;
        do v = next() & mask; while (v > rng);
        return v;
    }
};
inline uint32_t mask_of(uint32_t rng) {
    uint32_t m = rng;
    m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
    return m;
}
}  // namespace

/* ``np.stack([rng.randint(h, size=n) for h in high])`` (h <= 1: a row of zeros, nothing drawn) on the MT19937 state
 * (mt_key[624], *mt_pos), both advanced in place.  out int64 [W][n].  0, or -2 for a bound beyond 32 bits, -3 bad arguments. */
int64_t egk_host_bounded_draws(uint32_t* mt_key, int32_t* mt_pos, const int64_t* high, int64_t W, int32_t n, int64_t* out) {
    if (!mt_key || !mt_pos || !high || !out || n < 0 || W < 0 || *mt_pos < 0 || *mt_pos > 624) return -3;
    MT mt{mt_key, *mt_pos};
    for (int64_t w = 0; w < W; ++w) {
        const int64_t h = high[w];
        int64_t* o = out + w * n;
        if (h <= 1) {
            for (int32_t i = 0; i < n; ++i) o[i] = 0;
            continue;
        }
        if (h - 1 > 0xFFFFFFFFLL) return -2;
        const uint32_t rng = (uint32_t)(h - 1), mask = mask_of(rng);
        for (int32_t i = 0; i < n; ++i) o[i] = (int64_t)mt.bounded(rng, mask);
    }
    *mt_pos = mt.pos;
    return 0;
}

/* The store rows of W action windows -- feature_store.window_rows for every window in ONE call: numpy's arithmetic operation by
 * operation (double(size) / n, i * step, + draw, clip, round-half-even, integer cast; linspace = floor(i * step); uniform
 * sampling = linspace + size / n / 2), numpy's slice clipping of [start, end) to the video, and -1 rows where the reference's
 * np.take raises (empty window, an index == size).  ``random``: draws from the MT19937 state (advanced in place; NULL allowed
 * otherwise).  out int64 [W][n].  0, or -2 for a bound beyond 32 bits, -3 bad arguments. */
static inline int64_t window_rows_one(MT& mt, int64_t first_row, int64_t vl, int64_t start, int64_t end, int32_t n, int32_t random,
                                      int64_t* o) {
    int64_t lo = start < 0 ? 0 : start, hi = end < 0 ? 0 : end;
    lo = lo > vl ? vl : lo;
    hi = hi > vl ? vl : hi;
    const int64_t size = hi > lo ? hi - lo : 0;
    bool bad = size == 0;
    if (!bad) {
        const double step = (double)size / (double)n;
        const int64_t avg = size / n;
        uint32_t rng = 0, mask = 0;
        if (random && avg > 1) {
            if (avg - 1 > 0xFFFFFFFFLL) return -2;
            rng = (uint32_t)(avg - 1);
            mask = mask_of(rng);
        }
        for (int32_t i = 0; i < n; ++i) {
            int64_t idx;
            if (random && avg > 0) {
                const uint32_t v = avg > 1 ? mt.bounded(rng, mask) : 0u;
                double x = (double)i * step + (double)v;
                x = x < 0.0 ? 0.0 : (x > (double)size ? (double)size : x);
                idx = (int64_t)nearbyint(x);
            } else {
                idx = (int64_t)floor((double)i * step);
                if (!random) idx += size / n / 2;
            }
            if (idx >= size) bad = true;
            o[i] = first_row + lo + idx;
        }
    }
    if (bad)
        for (int32_t i = 0; i < n; ++i) o[i] = -1;
    return 0;
}

int64_t egk_host_window_rows(uint32_t* mt_key, int32_t* mt_pos, const int64_t* first_row, const int64_t* video_len,
                             const int64_t* start, const int64_t* end, int64_t W, int32_t n, int32_t random, int64_t* out) {
    if (!first_row || !video_len || !start || !end || !out || n <= 0 || W < 0) return -3;
    if (random && (!mt_key || !mt_pos || *mt_pos < 0 || *mt_pos > 624)) return -3;
    MT mt{mt_key, random ? *mt_pos : 0};
    for (int64_t w = 0; w < W; ++w) {
        const int64_t rc = window_rows_one(mt, first_row[w], video_len[w], start[w], end[w], n, random, out + w * n);
        if (rc) return rc;
    }
    if (random) *mt_pos = mt.pos;
    return 0;
}

/* One whole batch of a resident dataset in ONE host call (egopack_amd.data.SyntheticResidentDataset.batch, field for field).  The
 * reference builds a batch sample by sample in Python -- Dataset.__getitem__ (data/base_dataset.py:128-155: np.take over sampled
 * frame indices), the transform's edge list, PyG's Batch.from_data_list (utils/dataloading.py:56-70); here the dataset's per-sample
 * tables (labels, positions, window bounds, graph TEMPLATES with their CSR arrays in both orientations) are gathered, offset and
 * concatenated by this function.  It touches no Python object: ctypes releases the interpreter lock for its duration.  Consumes the
 * MT19937 stream exactly as the per-window calls do.  Output arrays are the caller's (sizes: egk_host_batch in the header). */
int64_t egk_host_build_batch(const egk_host_dataset* d, uint32_t* mt_key, int32_t* mt_pos, const int64_t* idx, int64_t B,
                             egk_host_batch* o) {
    if (!d || !idx || !o || B < 0 || d->T <= 0 || d->S <= 0 || d->y_elems < 0) return -3;
    if (d->train && (!mt_key || !mt_pos || *mt_pos < 0 || *mt_pos > 624)) return -3;
    const int64_t T = d->T, n = B * T;
    for (int64_t b = 0; b < B; ++b)
        if (idx[b] < 0 || idx[b] >= d->L || d->tau[idx[b]] < 0 || d->tau[idx[b]] >= d->n_tmpl) return -4;
    // ---- store rows of the B * T action windows (window by window, in batch order: the stream position of the per-sample calls)
    MT mt{mt_key, d->train ? *mt_pos : 0};
    for (int64_t b = 0; b < B; ++b) {
        const int64_t s = idx[b];
        for (int64_t t = 0; t < T; ++t) {
            const int64_t rc = window_rows_one(mt, d->first[s], d->vlen[s], d->starts[s * T + t], d->ends[s * T + t], d->S, d->train,
                                               o->x_idx + (b * T + t) * d->S);
            if (rc) return rc;
        }
    }
    if (d->train) *mt_pos = mt.pos;
    // ---- labels, positions, sequence ids
    int64_t pmin = 0, pmax = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t s = idx[b];
        for (int64_t e = 0; e < d->y_elems; ++e) o->y[b * d->y_elems + e] = d->y[s * d->y_elems + e];
        for (int64_t t = 0; t < T; ++t) {
            const int64_t p = d->pos[s * T + t];
            o->pos[b * T + t] = p;
            o->batch[b * T + t] = b;
            if ((b | t) == 0 || p < pmin) pmin = p;
            if ((b | t) == 0 || p > pmax) pmax = p;
        }
        o->ptr[b] = b * T;
        o->ptr32[b] = (int32_t)(b * T);
    }
    o->ptr[B] = n;
    o->ptr32[B] = (int32_t)n;
    o->pos_min = pmin;
    o->pos_max = pmax;
    // ---- the batch graph: the samples' templates, node ids shifted by b * T, edge ids by the edges in front (data.GraphTemplates.assemble)
    int64_t E = 0, nh = 0, nth = 0, dmax = 0, tdmax = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t k = d->tau[idx[b]];
        E += d->t_e[k];
        nh += d->t_nh[k];
        nth += d->t_nth[k];
        dmax = d->t_dmax[k] > dmax ? d->t_dmax[k] : dmax;
        tdmax = d->t_tdmax[k] > tdmax ? d->t_tdmax[k] : tdmax;
    }
    if (E != o->E || nh > o->heavy_cap || nth > o->t_heavy_cap) return -5;  // (the caller sized the arrays from the same tables)
    if (o->edge_cap && o->edge_cap < E) return -5;
    const int64_t ES = o->edge_cap ? o->edge_cap : E;  // row stride of edge_index
    int64_t eoff = 0, hh = 0, th = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t k = d->tau[idx[b]], e = d->t_e[k], noff = b * T;
        const int64_t* ei = d->t_ei + k * 2 * d->e_max;
        for (int64_t p = 0; p < e; ++p) {
            o->edge_index[eoff + p] = ei[p] + noff;
            o->edge_index[ES + eoff + p] = ei[d->e_max + p] + noff;
            o->col[eoff + p] = (int32_t)(d->t_col[k * d->e_max + p] + noff);
            o->t_col[eoff + p] = (int32_t)(d->t_tcol[k * d->e_max + p] + noff);
            o->t_wgt[eoff + p] = d->t_tw[k * d->e_max + p];
        }
        for (int64_t i = 0; i < T; ++i) {
            o->rowptr[noff + i] = (int32_t)(d->t_rp[k * (T + 1) + i] + eoff);
            o->t_rowptr[noff + i] = (int32_t)(d->t_trp[k * (T + 1) + i] + eoff);
            o->band[noff + i] = d->t_band[k * T + i];
        }
        for (int64_t q = 0; q < d->t_nh[k]; ++q) o->heavy[hh++] = (int32_t)(d->t_hv[k * d->h_max + q] + noff);
        for (int64_t q = 0; q < d->t_nth[k]; ++q) o->t_heavy[th++] = (int32_t)(d->t_thv[k * d->th_max + q] + noff);
        eoff += e;
    }
    for (int64_t p = E; p < ES; ++p) {  // (entries of the capacity regions no row range reaches)
        o->edge_index[p] = 0;
        o->edge_index[ES + p] = 0;
        o->col[p] = 0;
        o->t_col[p] = 0;
        o->t_wgt[p] = 0.0f;
    }
    o->rowptr[n] = (int32_t)E;
    o->t_rowptr[n] = (int32_t)E;
    o->n_heavy = nh;
    o->n_t_heavy = nth;
    o->heavy_mode = nh ? (dmax <= d->heavy_in_launch ? 1 : 0) : 0;
    o->t_heavy_mode = nth ? (tdmax <= d->heavy_in_launch ? 1 : 0) : 0;
    // ---- labelled rows of a per-node multi-head label tensor (data.live_label_rows): rows with a label in any head, in node order
    o->n_live = -1;
    o->live_ap_first = o->live_ap_step = 0;
    if (d->y_heads > 0 && d->y_elems == T * d->y_heads && o->live_idx && o->live_inv && o->live_y) {
        const int64_t H = d->y_heads;
        int64_t cnt = 0;
        for (int64_t r = 0; r < n; ++r) {
            bool any = false;
            for (int64_t h = 0; h < H; ++h) any = any || o->y[r * H + h] != -1;
            o->live_inv[r] = any ? cnt : -1;
            if (any) o->live_idx[cnt++] = r;
        }
        if ((double)cnt <= d->live_share * (double)n) {
            int64_t cap = (cnt + 63) / 64 * 64;
            cap = cap < 64 ? 64 : cap;
            if (cap > o->live_cap) return -5;
            for (int64_t q = 0; q < cap; ++q) {
                const int64_t r = q < cnt ? o->live_idx[q] : -1;
                if (q >= cnt) o->live_idx[q] = -1;
                for (int64_t h = 0; h < H; ++h) o->live_y[q * H + h] = r >= 0 ? o->y[r * H + h] : -1;
            }
            o->n_live = cnt;
            if (cap >= 2 && cnt == cap) {  // data.live_rows_progression: the padded list is full and evenly spaced
                const int64_t step = o->live_idx[1] - o->live_idx[0];
                bool ap = step >= 1;
                for (int64_t q = 1; ap && q < cap; ++q) ap = o->live_idx[q] - o->live_idx[q - 1] == step;
                if (ap) {
                    o->live_ap_first = o->live_idx[0];
                    o->live_ap_step = step;
                }
            }
        }
    }
    return 0;
}

int64_t egk_host_batch_sizes(const egk_host_dataset* d, const int64_t* idx, int64_t B, int64_t* out4) {
    if (!d || !idx || !out4 || B < 0 || d->T <= 0) return -3;
    const int64_t T = d->T, n = B * T;
    int64_t E = 0, nh = 0, nth = 0, live = -1;
    for (int64_t b = 0; b < B; ++b) {
        if (idx[b] < 0 || idx[b] >= d->L || d->tau[idx[b]] < 0 || d->tau[idx[b]] >= d->n_tmpl) return -4;
        const int64_t k = d->tau[idx[b]];
        E += d->t_e[k];
        nh += d->t_nh[k];
        nth += d->t_nth[k];
    }
    if (d->y_heads > 0 && d->y_elems == T * d->y_heads) {
        const int64_t H = d->y_heads;
        int64_t cnt = 0;
        for (int64_t b = 0; b < B; ++b) {
            const int64_t* y = d->y + idx[b] * d->y_elems;
            for (int64_t t = 0; t < T; ++t) {
                bool any = false;
                for (int64_t h = 0; h < H; ++h) any = any || y[t * H + h] != -1;
                cnt += any;
            }
        }
        if ((double)cnt <= d->live_share * (double)n) live = cnt;
    }
    out4[0] = E;
    out4[1] = nh;
    out4[2] = nth;
    out4[3] = live;
    return 0;
}

/* data.merge_batches / concat_csr in one host call: see the header. */
int64_t egk_host_merge_batches(const egk_host_part* parts, int32_t count, egk_host_merged* o) {
    if (!parts || !o || count <= 0) return -3;
    int64_t N = 0, E = 0;
    for (int32_t i = 0; i < count; ++i) {
        if (parts[i].n_nodes < 0 || parts[i].E < 0 || parts[i].edge_stride < parts[i].E) return -3;
        N += parts[i].n_nodes;
        E += parts[i].E;
    }
    if (o->edge_cap && o->edge_cap < E) return -5;
    const int64_t ES = o->edge_cap ? o->edge_cap : E;
    int64_t noff = 0, eoff = 0, hh = 0, th = 0, min_rows = 0;
    bool h_rows = false, h_all1 = true, th_rows = false, th_all1 = true;
    for (int32_t i = 0; i < count; ++i) {
        const egk_host_part& p = parts[i];
        for (int64_t r = 0; r < p.n_nodes; ++r) {
            o->pos[noff + r] = p.pos[r];
            o->rowptr[noff + r] = (int32_t)(p.rowptr[r] + eoff);
            o->t_rowptr[noff + r] = (int32_t)(p.t_rowptr[r] + eoff);
            o->band[noff + r] = p.band[r];
        }
        for (int64_t e = 0; e < p.E; ++e) {
            o->edge_index[eoff + e] = p.edge_index[e] + noff;
            o->edge_index[ES + eoff + e] = p.edge_index[p.edge_stride + e] + noff;
            o->col[eoff + e] = (int32_t)(p.col[e] + noff);
            o->t_col[eoff + e] = (int32_t)(p.t_col[e] + noff);
            o->t_wgt[eoff + e] = p.t_wgt[e];
        }
        for (int64_t q = 0; q < p.n_heavy; ++q) o->heavy[hh++] = (int32_t)(p.heavy[q] + noff);
        for (int64_t q = 0; q < p.n_t_heavy; ++q) o->t_heavy[th++] = (int32_t)(p.t_heavy[q] + noff);
        if (p.n_heavy) {
            h_rows = true;
            h_all1 = h_all1 && p.heavy_mode == 1;
        }
        if (p.n_t_heavy) {
            th_rows = true;
            th_all1 = th_all1 && p.t_heavy_mode == 1;
        }
        o->seg_ptr[i] = (int32_t)noff;
        if (i == 0 || p.n_nodes < min_rows) min_rows = p.n_nodes;
        if (i == 0 || p.pos_min < o->pos_min) o->pos_min = p.pos_min;
        if (i == 0 || p.pos_max > o->pos_max) o->pos_max = p.pos_max;
        noff += p.n_nodes;
        eoff += p.E;
    }
    for (int64_t e = E; e < ES; ++e) {
        o->edge_index[e] = 0;
        o->edge_index[ES + e] = 0;
        o->col[e] = 0;
        o->t_col[e] = 0;
        o->t_wgt[e] = 0.0f;
    }
    o->seg_ptr[count] = (int32_t)N;
    o->rowptr[N] = (int32_t)E;
    o->t_rowptr[N] = (int32_t)E;
    o->n_nodes = N;
    o->E = E;
    o->min_seg_rows = min_rows;
    o->heavy_mode = h_rows && h_all1 ? 1 : 0;
    o->t_heavy_mode = th_rows && th_all1 ? 1 : 0;
    return 0;
}

int egk_version(void) { return 100; }
const char* egk_last_error(void) { return g_err; }

int egk_tee_split_next(void* hi, void* lo, int64_t ld) {
    EGK_REQUIRE((hi && lo && ld > 0) || (!hi && !lo), "egk_tee_split_next: both halves (8-byte aligned) and a row stride, or NULL, NULL to disarm");
    EGK_REQUIRE(((reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 7) == 0, "egk_tee_split_next: halves must be 8-byte aligned");
    egk::arm_split_tee((egk::bf16_t*)hi, (egk::bf16_t*)lo, (long long)ld);
    return 0;
}

int egk_slab_input_next(const float* x2, const float* bias, float* x_out) {
    EGK_REQUIRE((x2 && x_out) || (!x2 && !bias && !x_out), "egk_slab_input_next: the second slab and the output, or NULLs to disarm");
    EGK_REQUIRE(((reinterpret_cast<uintptr_t>(x2) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(x_out)) & 15) == 0,
                "egk_slab_input_next: 16-byte aligned pointers");
    egk::arm_slab_input(x2, bias, x_out);
    return 0;
}

int egk_gemm_defer_reduce_next(int32_t on) {
    egk::arm_defer_reduce(on != 0);
    return 0;
}

int egk_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return 0;
}
int egk_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain();
    for (auto& t : g_tot) t = Totals();
    return 0;
}
int egk_prof_count(void) { return KID_COUNT; }
int egk_prof_get(int id, char* name, int name_len, int64_t* launches, double* total_ms, double* alg_flops,
                 double* alg_bytes) {
    if (id < 0 || id >= KID_COUNT) return EGK_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    drain();
    if (name && name_len > 0) {
        strncpy(name, kNames[id], name_len - 1);
        name[name_len - 1] = 0;
    }
    if (launches) *launches = g_tot[id].launches;
    if (total_ms) *total_ms = g_tot[id].ms;
    if (alg_flops) *alg_flops = g_tot[id].flops;
    if (alg_bytes) *alg_bytes = g_tot[id].bytes;
    return 0;
}
}
