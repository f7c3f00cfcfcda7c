/*
 * egopack_hip.h -- C ABI of libegopack_hip.so: the MI355X (gfx950) kernels behind the
 * EgoPack training hot path.
 *
 * Boundary rules (last row):
 *   - plain C: raw device pointers, sizes, an opaque stream handle; no torch / C++ types;
 *   - every launcher is stream-ordered, re-entrant, allocates nothing, never synchronises
 *     (so a caller may capture it into a hipGraph); the caller owns every buffer;
 *   - return value: 0 = ok, negative = EGK_E* argument error, positive = hipError_t;
 *     egk_last_error() returns a thread-local message.  No exception crosses the ABI.
 *   - all matrices are row-major; "ld*" = leading dimension in ELEMENTS.
 *
 * Each entry point names the reference interface it replaces (path:line relative to the
 * reference repository sapeirone/EgoPack).  The reference has no FFI of its own (it is pure
 * Python on torch/torch_geometric); the binding a maintainer adds is the ctypes loader shown
 * in INTEGRATION.md (egopack_amd/_lib.py is that loader).
 */
#ifndef EGOPACK_HIP_H
#define EGOPACK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* egk_stream_t; /* a hipStream_t */

enum { EGK_OK = 0, EGK_EINVAL = -1, EGK_EUNSUPPORTED = -2 };
enum { EGK_F32 = 0, EGK_BF16 = 1 };                 /* element type of a matrix in memory      */
enum { EGK_COMPUTE_F32 = 0, EGK_COMPUTE_BF16 = 1 }; /* MFMA type: exact f32 16x16x4 | bf16 16x16x32 (f32 accumulate) */
enum { EGK_ACT_NONE = 0, EGK_ACT_RELU = 1 };

/* ---- library ------------------------------------------------------------------------- */
int egk_version(void);
const char* egk_last_error(void);

/* Built-in per-kernel timing with HIP events recorded on the launch stream (bench.py's
 * roofline leg).  Enable only outside graph capture.  egk_prof_get() synchronises the
 * recorded events and returns the totals for kernel id in [0, egk_prof_count()). */
int egk_prof_enable(int on);
int egk_prof_reset(void);
int egk_prof_count(void);
int egk_prof_get(int id, char* name, int name_len, int64_t* launches, double* total_ms,
                 double* alg_flops, double* alg_bytes);

/* Development aid: a one-lane launch that writes the device wall clock (100 MHz constant-rate counter) to buf[idx] when
 * the stream reaches it -- phase boundaries of a captured step timed in place (tools/phase_stamps.py). */
int egk_stamp(egk_stream_t s, uint64_t* buf, int32_t idx);


/* One-shot "split tee" for the three-product contractions (egk_split_bf16's halves without its launch): the NEXT call, on this
 * host thread, of egk_rowln_fwd, egk_graphln_fwd, egk_graphln_fwd_apply, egk_pe_add, egk_pe_add_table, egk_csr_gather or
 * egk_csr_gather_banded -- with an f32 result [rows, cols] -- also stores hi = bf16(result) and lo = bf16(result - hi) to
 * hi / lo + row * ld (8-byte aligned, ld >= cols), then the tee is disarmed.  The producers of every contraction operand of the
 * forward-only precise pass (egopack_amd.engine.EgoPackStep.precise_aux_features; reference: the f32 activations of
 * models/graph.py:53-63 feeding models/graphONE/graphONE.py:119-141) are exactly these row kernels.  NULL, NULL disarms. */
int egk_tee_split_next(void* hi, void* lo, int64_t ld);

/* Split-K contractions of the forward-only precise pass WITHOUT their reduce launch (11 such launches of ~6 us sat on that pass's
 * chain in BASELINE config 4, each in front of a row kernel that reads the reduced matrix once):
 *   egk_gemm_defer_reduce_next(1): the NEXT egk_gemm call of this host thread, if it splits K (bf16-operand pipelined launch, f32
 *     result, no accumulate / activation / residual / alpha), leaves its slabs in the caller's workspace ws = [splitk][M][N] and
 *     does not launch gemm_splitk_reduce; the bias is not applied either.
 *   egk_slab_input_next(x2, bias, x_out): the NEXT call of egk_rowln_fwd, egk_graphln_fwd or egk_pe_add_table (f32 rows of <= 1024
 *     columns, a multiple of 4) reads its input as (x[r, c] + x2[r, c]) + bias[c] -- the reduce launch's arithmetic, the same bits --
 *     and stores that value to x_out[r, c] for later readers (reference: the Linear in front of nn.LayerNorm / gnn.LayerNorm /
 *     PositionalEncoding in models/temporal_pooling/trn_pooling.py:28-45, models/graph.py:39-63).  NULLs disarm.
 *   egk_gemm_reduce_slabs: the reduce launch by itself, for slabs nobody consumed that way. */
int egk_gemm_defer_reduce_next(int32_t on);
int egk_slab_input_next(const float* x2, const float* bias, float* x_out);
int egk_gemm_reduce_slabs(egk_stream_t s, const float* ws, int32_t splitk, int32_t M, int32_t N, const float* bias, float* C, int64_t ldc);

/* ---- dense contractions (MFMA) ---------------------------------------------------------
 * C[M,N] = act(alpha * (op(A) . op(B)^T) + (accumulate ? C : 0) + bias[n]) + residual[m,n]
 * where the contraction runs over K = K1 + K2 with a two-source split:
 *   k <  K1 : A1 / B1      k >= K1 : A2 / B2     (K2 = 0 -> single source)
 * op(A)[m,k] = transA ? A[k*lda + m] : A[m*lda + k];  op(B)[n,k] = transB ? B[k*ldb + n] : B[n*ldb + k].
 * Replaces every torch.nn.Linear / gnn.Linear forward and its autograd backward on the path:
 *   models/temporal_pooling/trn_pooling.py:30-40 (TRN MLP), models/graph.py:42,46 (SAGEConv
 *   lin / [lin_l|lin_r] as ONE two-source contraction, final Linear + residual),
 *   models/tasks/task.py:17-23, recognition.py:42, lta.py:42, oscc.py:71, pnr.py:64
 *   (projections and classifiers), models/graphONE/graphONE.py:60-63,148-151 (stage linears,
 *   cosine-similarity product).
 */
typedef struct egk_gemm_desc {
    int32_t M, N, K1, K2;
    const void *A1, *A2;
    const void *B1, *B2;
    int64_t lda1, lda2, ldb1, ldb2;
    int32_t transA, transB;
    int32_t a_dtype, b_dtype; /* element type in memory: EGK_F32 or EGK_BF16 (same for A and B; bf16 needs EGK_COMPUTE_BF16) */
    int32_t compute;          /* EGK_COMPUTE_* */
    void* C;
    int64_t ldc;
    int32_t c_dtype;          /* EGK_F32 or EGK_BF16 (accumulate needs EGK_F32) */
    int32_t accumulate;
    int32_t act;
    float alpha;
    const float* bias;     /* [N] or NULL */
    const void* residual;  /* [M, ldr] of r_dtype, or NULL */
    int64_t ldr;
    int32_t r_dtype;
    /* split-K: splitk > 1 writes f32 partial slabs [splitk][M][N] to ws (plain stores) and a
     * second launch sums them in slab order and applies the epilogue (bitwise reproducible, no
     * atomics).  ws_bytes >= splitk*M*N*4.  egk_gemm_splitk() is the library's policy. */
    int32_t splitk;
    void* ws;
    int64_t ws_bytes;
    /* dW form only (transA = 1, K2 = 0): dbias[m] += sum_k op(A)[m, k], i.e. the bias gradient colsum(dY) of the
     * Linear whose weight gradient this launch computes -- summed from the dY^T image the kernel already holds in
     * LDS (or by explicit column-sum launches when the pipelined kernel is not eligible).  Needs
     * ws_bytes >= egk_gemm_ws_bytes(desc). */
    float* dbias;
    /* Per-segment sums of the RESULT for the graph-mode LayerNorm that consumes it (gnn.LayerNorm(mode='graph') at
     * models/graph.py:43 -- statistics over all elements of a task batch = a row segment), taken in the epilogue so that the
     * LayerNorm's own statistics pass over the tensor disappears:
     *   st_mode 1  (sum c, sum c^2) of the stored result c                      -> egk_graphln_fwd_apply
     *   st_mode 2  the result is dy of that LayerNorm + LeakyReLU: (sum dxhat, sum dxhat * xhat), dxhat = dy * lrelu'(pre) * w,
     *              xhat = (x - mean) * rinv, pre = xhat * w + b                  -> egk_graphln_bwd_apply
     * per row segment [st_seg_ptr[s], st_seg_ptr[s+1]) (device int32 [st_nseg + 1]; st_min_seg_rows = the shortest segment,
     * for the host-side check that a tile spans at most two), one partial per output tile:
     *   st_ws = double [egk_gemm_stats_blocks(desc)][st_nseg][2].
     * Only some tile variants can do it: ask egk_gemm_stats_blocks() first (0 = run the LayerNorm's statistics pass). */
    int32_t st_mode, st_nseg, st_min_seg_rows;
    const int32_t* st_seg_ptr;
    void* st_ws;
    const void* st_x;      /* st_mode 2: the LayerNorm's input, laid out like C (leading dimension st_ldx, element type c_dtype) */
    int64_t st_ldx;
    const float* st_stats; /* st_mode 2: [st_nseg][2] (mean, 1 / (std + eps)) from the forward pass */
    const float* st_w;
    const float* st_b;
    float st_slope;
    /* Extra K sources beyond (A1, B1, K1) and (A2, B2, K2): C = epilogue(sum over ALL sources of op(A_s) . op(B_s)^T), same
     * layout (transA / transB) and element type for every source, walked in the order K1, K2, xK[0 .. n_extra-1].  Up to 4
     * (six sources in all).  What they are for: a contraction of f32 values at (close to) f32 precision on the bf16 matrix
     * pipe -- each f32 operand is split into bf16 halves x = hi + lo (egk_split_bf16) and a.b is taken as
     * a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, three sources per logical one, f32 accumulation throughout (the forward-only,
     * higher-precision feature path in front of the nearest-prototype index op, models/graphONE/graphONE.py:119-141). */
    int32_t n_extra;
    int32_t xK[4];
    const void* xA[4];
    const void* xB[4];
    int64_t xlda[4], xldb[4];
    /* 1: the 16-bit operands (a_dtype = b_dtype = EGK_BF16 as the storage width) hold IEEE half values and the launch multiplies
     * them on the f16 matrix instructions (f32 accumulation): 11 significand bits instead of bf16's 8 -- the one-product screen of
     * the nearest-prototype search (egk_topk_window_group, screen_f16).  egk_gemm_grouped only, row-major A and B, no split-K /
     * statistics epilogue; every problem of the launch must agree. */
    int32_t op_f16;
} egk_gemm_desc;
/* workspace bytes a descriptor needs (split-K slabs + bias-gradient partials / column-sum scratch) */
int64_t egk_gemm_ws_bytes(const egk_gemm_desc* d);
int egk_gemm(egk_stream_t s, const egk_gemm_desc* d);
int egk_gemm_stats_blocks(const egk_gemm_desc* d);
/* HOST helper of the batch builders (runs on the CPU, touches no device): ``np.stack([rng.randint(h, size = n) for h in
 * high])`` of numpy's legacy RandomState on the generator's own state -- MT19937, mt_key[624] + *mt_pos as
 * ``RandomState.get_state()`` returns them, both advanced in place (per value: v = next_uint32 & mask until v <= high - 1; a
 * bound <= 1 consumes nothing) -- the consumption pattern of the reference's per-window segment sampling
 * (data/base_dataset.py:128-139), so that a whole batch of windows is sampled in one call on the SAME random stream.
 * out int64 [W][n].  Returns 0, -2 for a bound beyond 32 bits, -3 for bad arguments. */
int64_t egk_host_bounded_draws(uint32_t* mt_key, int32_t* mt_pos, const int64_t* high, int64_t W, int32_t n, int64_t* out);

/* HOST helper: the feature-store rows of W action windows in one call -- the reference's per-window segment sampling
 * (BaseFrameDataset.random_sampling_indices / uniform_sampling_indices, data/base_dataset.py:128-155, applied to
 * video_features[start:end] and np.take'n as in data/ego4d_fho.py:228-236), numpy's arithmetic operation by operation, random
 * offsets drawn from the MT19937 state as egk_host_bounded_draws does (mt_key / mt_pos may be NULL when ``random`` is 0); a
 * window the reference replaces by an all-zero clip gives n times -1.  out int64 [W][n].  Returns 0, -2, -3 as above. */
int64_t egk_host_window_rows(uint32_t* mt_key, int32_t* mt_pos, const int64_t* first_row, const int64_t* video_len,
                             const int64_t* start, const int64_t* end, int64_t W, int32_t n, int32_t random, int64_t* out);

/* A whole batch of a device-resident dataset in ONE host call (egopack_amd.data.SyntheticResidentDataset.batch field for field;
 * reference: Dataset.__getitem__ data/base_dataset.py:128-155 + the transform's edges + PyG Batch.from_data_list,
 * utils/dataloading.py:56-70, done sample by sample in Python).  ``egk_host_dataset``: the dataset's per-sample tables, built once
 * (they are functions of its seed); ``egk_host_batch``: caller-allocated outputs.  Holds no Python object (ctypes drops the
 * interpreter lock for the call).  0, -2 / -3 as egk_host_window_rows, -4 a sample or template index out of range, -5 an output
 * array sized for another batch. */
typedef struct {
    int32_t T, S;                /* nodes per sample, segments per node */
    int32_t train;               /* random segment sampling (consumes the MT19937 stream) */
    int32_t y_heads;             /* > 0: labels are [T, y_heads] per sample (AR / LTA): the labelled-row lists are built too */
    int64_t L, y_elems;          /* samples; int64 label elements per sample */
    int64_t heavy_in_launch;     /* data.HEAVY_IN_LAUNCH_DEGREE */
    double live_share;           /* data.LIVE_ROWS_MAX_SHARE */
    const int64_t* y;            /* [L][y_elems] */
    const int64_t* pos;          /* [L][T] */
    const int64_t* tau;          /* [L] graph template of each sample */
    const int64_t* first;        /* [L] first store row of the sample's video */
    const int64_t* vlen;         /* [L] frames of that video */
    const int64_t* starts;       /* [L][T] window bounds */
    const int64_t* ends;
    int64_t n_tmpl, e_max, h_max, th_max;
    const int64_t* t_e;          /* [n_tmpl] edges */
    const int64_t* t_ei;         /* [n_tmpl][2][e_max] edge lists (source row, target row) */
    const int64_t* t_col;        /* [n_tmpl][e_max] CSR by target */
    const int64_t* t_tcol;       /* [n_tmpl][e_max] CSR by source */
    const float* t_tw;           /* [n_tmpl][e_max] 1 / in-degree of the target */
    const int64_t* t_rp;         /* [n_tmpl][T + 1] */
    const int64_t* t_trp;
    const uint8_t* t_band;       /* [n_tmpl][T] */
    const int64_t* t_nh;         /* [n_tmpl] listed heavy rows per orientation */
    const int64_t* t_nth;
    const int64_t* t_hv;         /* [n_tmpl][h_max] */
    const int64_t* t_thv;        /* [n_tmpl][th_max] */
    const int64_t* t_dmax;       /* [n_tmpl] largest degree per orientation */
    const int64_t* t_tdmax;
} egk_host_dataset;
typedef struct {
    int64_t E;                   /* in: edges of this batch (sum of t_e over its samples: the arrays below are sized by it) */
    int64_t heavy_cap, t_heavy_cap, live_cap;   /* in: capacities of heavy / t_heavy / live_idx, live_y */
    int64_t* y;                  /* [B][y_elems] */
    int64_t* pos;                /* [B * T] */
    int64_t* batch;              /* [B * T] */
    int64_t* ptr;                /* [B + 1] */
    int32_t* ptr32;              /* [B + 1] */
    int64_t* x_idx;              /* [B * T][S] store rows (-1: an all-zero clip) */
    int64_t* edge_index;         /* [2][E] */
    int32_t* rowptr;             /* [B * T + 1] */
    int32_t* col;                /* [E] */
    int32_t* t_rowptr;
    int32_t* t_col;
    float* t_wgt;                /* [E] */
    uint8_t* band;               /* [B * T] */
    int32_t* heavy;              /* [heavy_cap] */
    int32_t* t_heavy;
    int64_t* live_idx;           /* [live_cap] or NULL */
    int64_t* live_inv;           /* [B * T] or NULL */
    int64_t* live_y;             /* [live_cap][y_heads] or NULL */
    /* out */
    int64_t n_heavy, n_t_heavy, n_live /* -1: not a compactable label tensor, or too many labelled rows */, pos_min, pos_max;
    int32_t heavy_mode, t_heavy_mode;
    /* in: 0 = edge_index is [2][E]; > 0 = the edge-sized arrays live in regions of this CAPACITY (>= E) of a packed transfer
     * buffer (data.EDGE_BUCKET): edge_index is [2][edge_cap] and entries E .. edge_cap of edge_index / col / t_col / t_wgt are
     * cleared, as data.to_device_packed does */
    int64_t edge_cap;
    /* out: the padded labelled-row list is the arithmetic progression first, first + step, ... (data.live_rows_progression);
     * step 0: it is not */
    int64_t live_ap_first, live_ap_step;
} egk_host_batch;
int64_t egk_host_build_batch(const egk_host_dataset* ds, uint32_t* mt_key, int32_t* mt_pos, const int64_t* idx, int64_t B,
                             egk_host_batch* out);
/* What the caller sizes a batch's arrays by, from the dataset's tables alone: out4 = {E, listed heavy rows, listed heavy rows of
 * the transposed orientation, labelled rows (-1 as n_live above)}.  0 / -3 / -4 as egk_host_build_batch. */
int64_t egk_host_batch_sizes(const egk_host_dataset* ds, const int64_t* idx, int64_t B, int64_t* out4);

/* The merged batch of a fused backbone pass from its task batches in ONE host call (egopack_amd.data.merge_batches / concat_csr
 * field for field; the reference runs the backbone per task batch, main_temporal.py:87-90): positions concatenated, edge lists and
 * both CSR orientations concatenated with node / edge offsets, the heavy-row lists shifted, segment boundaries.  ``parts``: the
 * task batches in order (pointers as egk_host_batch filled them; edge_stride = the row stride of their edge_index). */
typedef struct {
    int64_t n_nodes, E, n_heavy, n_t_heavy, edge_stride, pos_min, pos_max;
    const int64_t* pos;
    const int64_t* edge_index;
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* t_rowptr;
    const int32_t* t_col;
    const float* t_wgt;
    const uint8_t* band;
    const int32_t* heavy;
    const int32_t* t_heavy;
    int32_t heavy_mode, t_heavy_mode;
} egk_host_part;
typedef struct {
    int64_t edge_cap;            /* in: as egk_host_batch.edge_cap (0: [2][sum E]) */
    int64_t* pos;                /* [sum n] */
    int64_t* edge_index;
    int32_t* rowptr;             /* [sum n + 1] */
    int32_t* col;
    int32_t* t_rowptr;
    int32_t* t_col;
    float* t_wgt;
    uint8_t* band;               /* [sum n] */
    int32_t* heavy;              /* [sum n_heavy] */
    int32_t* t_heavy;
    int32_t* seg_ptr;            /* [count + 1] */
    /* out */
    int64_t n_nodes, E, min_seg_rows, pos_min, pos_max;
    int32_t heavy_mode, t_heavy_mode;
} egk_host_merged;
int64_t egk_host_merge_batches(const egk_host_part* parts, int32_t count, egk_host_merged* out);

/* x = hi + lo with hi = bf16(x) (round to nearest even) and lo = bf16(x - hi): the two bf16 operands that stand for an f32
 * matrix in a three-product contraction (egk_gemm_desc extra sources).  src f32 [rows, cols] with leading dimension ld_src;
 * hi (may be NULL: only lo is wanted, e.g. when hi is the bf16 copy the optimizer already keeps) and lo bf16 [rows, cols]
 * with leading dimension ld_out. */
int egk_split_bf16(egk_stream_t s, const float* src, int64_t ld_src, void* hi, void* lo, int64_t ld_out, int64_t rows,
                   int64_t cols);

/* Grouped launch: ``count`` (<= 8) independent contractions of the SAME layout (transA / transB), bf16 operands with
 * 16-byte aligned rows, every K source a multiple of 64, no split-K, in ONE launch (blockIdx.y = problem).  Replaces the
 * per-task projection heads of the multi-task step -- ProjectionTask.forward_features of every enabled task
 * (models/tasks/task.py:17-26, called per task at main_temporal.py:93-126) -- whose contractions have the same shapes but
 * read different rows of the backbone output and different weights.  Results are bit-identical to ``count`` egk_gemm
 * calls that pick the same tile variant.  dbias (fused bias gradient of the dW form) is allowed. */
int egk_gemm_grouped(egk_stream_t s, const egk_gemm_desc* descs, int32_t count);
int egk_gemm_splitk(int32_t M, int32_t N, int32_t K, int32_t compute);
/* development knob of the row kernels (launch geometry only, results unchanged): returns the previous value */
int egk_tune(int32_t key, int32_t value);
/* development knob for A/B measurements in one process (results unchanged up to the summation order of variant 5):
 * 0 routes every contraction through the generic register-staged kernel; 1 (default) lets eligible ones (bf16
 * operands, 16-byte aligned rows, K % 64 == 0) use the LDS-DMA pipelined kernels, variant chosen per launch; a value
 * 2..11 forces one variant where it is legal (2 / 3 / 4: 3- / 2- / 4-stage ring on 128 x 128 tiles, 5: two wave groups,
 * 6: 256 x 128 tile, 7: 256 x 256 tile, 8 / 11: 96- / 64-row tiles); 100 + g overrides the XCD tile-group height
 * (100 = policy); 200 / 201 selects the direct / row-contiguous (default) epilogue of the 4-wave variants; 300 / 301 the
 * spread / XCD-packed (default) workgroup placement of grouped launches.  Returns the previous variant setting. */
int egk_gemm_set_pipeline(int32_t on);

/* out[n] (+)= sum_m x[m, n] : bias gradients of every Linear above.  Two launches (row-chunk
 * partials in ws, then a fixed-order sum); ws: float[egk_colsum_ws_len(M, N)]. */
int egk_colsum_ws_len(int32_t M, int32_t N);
int egk_colsum(egk_stream_t s, const void* x, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate,
               float* ws, int32_t dtype);

/* ---- row LayerNorm (+ReLU, +dropout)  nn.LayerNorm -> ReLU -> Dropout -----------------
 * trn_pooling.py:31-33,36-38; task.py:19-20; graphONE.py:61-62.
 * y = dropout(relu((x-mean_row)*rstd_row*w + b)); saves mean/rstd per row; keep mask (u8, 1=keep)
 * is written when p>0 (Philox4x32-10 keyed by seed, counter = element index + offset).
 * relu=0 disables the activation. mask may be NULL iff p==0.
 * dev_offset (device, uint64[1], may be NULL) is added to ``offset`` inside the kernel: a caller
 * that replays a captured graph advances it between replays so every step draws a fresh mask. */
int egk_rowln_fwd(egk_stream_t s, const void* x, const float* w, const float* b, void* y, float* mean,
                  float* rstd, uint8_t* mask, int32_t rows, int32_t cols, float eps, int32_t relu, float p,
                  uint64_t seed, uint64_t offset, const uint64_t* dev_offset, int32_t dtype);
/* dx; dw/db are ACCUMULATED into dw[cols], db[cols] through per-block partials in ws
 * (ws: float[2 * egk_rowln_bwd_ws_rows(rows) * cols]). */
int egk_rowln_bwd_ws_rows(int32_t rows);
int egk_rowln_bwd(egk_stream_t s, const void* dy, const void* x, const float* w, const float* b, const float* mean,
                  const float* rstd, const uint8_t* mask, void* dx, float* dw, float* db, float* ws, int32_t rows,
                  int32_t cols, int32_t relu, float p, int32_t dtype);

/* ---- graph-mode LayerNorm + LeakyReLU  gnn.LayerNorm(mode='graph', batch=None) -> LeakyReLU
 * models/graph.py:43-44.  Statistics span all rows*cols elements of one SEGMENT of rows; seg_ptr
 * [n_seg+1] gives row ranges (one segment per task batch: the reference runs one backbone pass
 * per task batch, main_temporal.py:87-90; here the passes are fused and the statistics stay
 * per task batch).  y = lrelu((x-mean_s)/(std_s+eps)*w+b, slope).
 * stats: float[n_seg*2] = (mean, 1/(std+eps)) written by fwd, read by bwd.
 * ws: egk_graphln_ws_bytes(rows, cols, n_seg) bytes of scratch, 16-byte aligned.
 * bwd ACCUMULATES dw/db (either may be NULL).  n_seg <= 16. */
int64_t egk_graphln_ws_bytes(int32_t rows, int32_t cols, int32_t n_seg);
int egk_graphln_fwd(egk_stream_t s, const void* x, const float* w, const float* b, void* y, float* stats,
                    const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols, float eps, float slope,
                    void* ws, int32_t dtype);
int egk_graphln_bwd(egk_stream_t s, const void* dy, const void* x, const float* w, const float* b,
                    const float* stats, void* dx, float* dw, float* db, const int32_t* seg_ptr, int32_t n_seg,
                    int32_t rows, int32_t cols, float eps, float slope, void* ws, int32_t dtype);
/* The second launches alone, with the per-segment sums taken from per-block partials computed ELSEWHERE -- by the
 * epilogue of the contraction that produced x (forward, egk_gemm_desc.st_mode 1) or dy (backward, st_mode 2):
 *   partials = double [n_partials][n_seg][2]  ((sum x, sum x^2)  /  (sum dxhat, sum dxhat * xhat) per segment).
 * bwd_apply also writes the partial rows of dw / db (the statistics pass used to): ws_col f32
 * [egk_rowln_bwd_ws_rows(rows)][2][cols], reduced by egk_ln_bwd_reduce(ws_col, dw, db, rows, cols, 0). */
int egk_graphln_fwd_apply(egk_stream_t s, const void* x, const float* w, const float* b, void* y, float* stats,
                          const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols, float eps, float slope,
                          const void* partials, int32_t n_partials, int32_t dtype);
int egk_graphln_bwd_apply(egk_stream_t s, const void* dy, const void* x, const float* w, const float* b, const float* stats,
                          void* dx, const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols, float eps, float slope,
                          const void* partials, int32_t n_partials, float* ws_col, int32_t dtype);
/* The passes of egk_graphln_fwd / egk_graphln_bwd as separate calls, so that a data-parallel run can combine the segment
 * sums of all ranks between the statistics pass and the normalising pass (exact global-batch statistics; the reference
 * has one process, so its gnn.LayerNorm(mode='graph') at models/graph.py:43 sees the whole batch):
 *   egk_graphln_stats       partials = double [egk_graphln_stats_blocks(rows)][n_seg][2]: per-block (sum x, sum x^2);
 *                           follow with egk_graphln_fwd_apply(partials, n_partials)
 *   egk_graphln_bwd_stats   head of ws = double [egk_graphln_stats_blocks(rows)][n_seg][2]: per-block
 *                           (sum dxhat, sum dxhat * xhat); the dw / db partial rows follow in ws (layout of egk_graphln_bwd)
 *   egk_graphln_bwd_finish  dx from the sums in ``partials`` (the head of ws, or combined sums) + the dw / db reduction of ws
 * The kernels divide the sums they are given by the LOCAL element count of a segment: sums combined over ranks are passed
 * scaled by local count / global count.  egk_graphln_bwd == bwd_stats + bwd_finish(partials = ws). */
int32_t egk_graphln_stats_blocks(int32_t rows);
int egk_graphln_stats(egk_stream_t s, const void* x, const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols,
                      void* partials, int32_t dtype);
int egk_graphln_bwd_stats(egk_stream_t s, const void* dy, const void* x, const float* w, const float* b, const float* stats,
                          const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols, float slope, void* ws,
                          int32_t dtype);
int egk_graphln_bwd_finish(egk_stream_t s, const void* dy, const void* x, const float* w, const float* b, const float* stats,
                           void* dx, float* dw, float* db, const int32_t* seg_ptr, int32_t n_seg, int32_t rows, int32_t cols,
                           float eps, float slope, const void* partials, int32_t n_partials, const void* ws, int32_t dtype);
/* The parameter-gradient reduction of egk_rowln_bwd (n_seg = 0) / egk_graphln_bwd (n_seg >= 1) as its own launch: call
 * those with dw = db = NULL (they then leave the per-workgroup partial rows in ws) and this one, with the same ws, rows,
 * cols, on whatever stream should carry it -- dw / db feed nothing but the optimizer, the kernels that need dx need not
 * wait for them.  dw[c] += sum of partials, db[c] += ... (accumulating, like the fused form). */
int egk_ln_bwd_reduce(egk_stream_t s, const void* ws, float* dw, float* db, int32_t rows, int32_t cols, int32_t n_seg);
/* ``count`` (<= 8) of those reductions in one launch (the norm layers whose backward ran since the last weight-gradient
 * flush); same summation order as egk_ln_bwd_reduce: same bits. */
int egk_ln_bwd_reduce_multi(egk_stream_t s, const void* const* ws, float* const* dw, float* const* db, const int32_t* rows,
                            const int32_t* cols, const int32_t* n_seg, int32_t count);
/* ---- one-logit classifier + BCE-with-logits in one pass over the rows: the PNR head
 * models/tasks/pnr.py:20,37-52 (Linear(features, 1) -> squeeze) + nn.BCEWithLogitsLoss(reduction='none')
 * (main_temporal.py:117-121).  A [rows, cols] x [cols, 1] contraction is a row reduction: per row n
 *   z = <f_n, w> + bias[0];  logits[n] = z;  loss[n] = BCE(z, y[n])  (the formula of egk_bce_fwd)
 * and, when df != NULL (the seed of the loss vector's backward is known: it is the constant weight / numel),
 *   g = (sigmoid(z) - y[n]) * seed rounded to the element type;  df_n = g w;  dw += sum_n g f_n;  db += sum_n g
 * f, w, df: element type ``dtype`` (w = the operand copy of the weight row the contraction path reads).
 * ws: float [egk_rowdot_ws_rows(rows)][cols + 4] partial rows of dw / db; egk_rowdot_reduce ACCUMULATES them into
 * dw [cols] / db [1] in block order (fixed order: bitwise reproducible). */
int32_t egk_rowdot_ws_rows(int32_t rows);
int egk_rowdot_bce(egk_stream_t s, const void* f, const void* w, const float* bias, const int64_t* y, float* logits,
                   float* loss, void* df, float* ws, int32_t rows, int32_t cols, float seed, int32_t dtype);
int egk_rowdot_reduce(egk_stream_t s, const float* ws, float* dw, float* db, int32_t rows, int32_t cols);
/* Two-logit classifier + cross entropy over FEW rows in one launch (the OSCC head of the multi-task loop: max-pooled sequence
 * features -> Linear(H, 2) -> nn.CrossEntropyLoss(reduction='none', ignore_index=-1), models/tasks/oscc.py:65-79 +
 * main_temporal.py:291, :99): logits [rows, 2] = f w^T + bias, loss [rows] as egk_ce_fwd (label smoothing ``smoothing``), and --
 * df != NULL -- from the announced backward seed: g = seed (softmax - target) rounded to the element type, df = g w,
 * dw [2, cols] += g^T f, db [2] += column sums of g (rows in order: bitwise reproducible).  f, w, df: element type ``dtype``
 * (w = the operand copy of the classifier's weight rows).  gws: float [rows][2] scratch (the rounded logit gradients between the
 * row launch and the column launch), required with gradients.  rows <= egk_rowdot_ce2_max_rows(). */
int32_t egk_rowdot_ce2_max_rows(void);
int egk_rowdot_ce2(egk_stream_t s, const void* f, const void* w, const float* bias, const int64_t* y, float* logits, float* loss,
                   void* df, float* dw, float* db, float* gws, int32_t rows, int32_t cols, float smoothing, float seed, int32_t dtype);
/* The same with n_src <= 4 sources, each with its own classifier: logits = scale * sum_k (f_k w_k^T + bias_k), scale = 1 / n_src
 * when ``average`` (the EgoPack head: the primary pooled features + one pooled GraphONE feature per auxiliary task, fused as
 * ``stack([...]).mean(0)`` / ``.sum(0)``; models/tasks/oscc.py:70-78, main_egopack.py:279), the loss of egk_rowdot_ce2 on them, and
 * per source df_k = g w_k, dw_k += g^T f_k, db_k += column sums of g with the ONE rounded g = scale * seed (softmax - target).
 * Pointer arrays of n_src entries (host memory; bias / df / dw / db arrays or entries may be NULL).  ``average`` bit 0: mean
 * instead of sum; bits 1-2: 0 = both launches, 1 = the row launch only (logits, loss, df, gws), 2 = the column launch only (dw, db
 * from the gws of an earlier phase-1 call with the same sources): a caller may issue the parameter gradients in its backward. */
int egk_rowdot_ce2_multi(egk_stream_t s, int32_t n_src, const void* const* f, const void* const* w, const float* const* bias,
                         const int64_t* y, float* logits, float* loss, void* const* df, float* const* dw, float* const* db,
                         float* gws, int32_t rows, int32_t cols, int32_t average, float smoothing, float seed, int32_t dtype);
/* Grouped row LayerNorm(+ReLU): n_groups (<= 4) consecutive row ranges [row_ptr[g], row_ptr[g+1]) of ONE [rows, cols]
 * matrix, each with its own (w, b) -- the LayerNorms of the per-task projection heads (models/tasks/task.py:20-21) in one
 * launch.  No dropout.  bwd writes dx and per-workgroup partial rows of dw / db:
 *   ws[(g * egk_rowln_bwd_ws_rows(max range rows) + block)][2][cols]   (f32)
 * reduce range g with egk_ln_bwd_reduce(ws + g * blocks * 2 * cols, dw_g, db_g, max range rows, cols, 0). */
int egk_rowln_group_fwd(egk_stream_t s, const void* x, const float* const* w, const float* const* b, const int32_t* row_ptr,
                        int32_t n_groups, void* y, float* mean, float* rstd, int32_t cols, float eps, int32_t relu,
                        int32_t dtype);
int egk_rowln_group_bwd(egk_stream_t s, const void* dy, const void* x, const float* const* w, const float* const* b,
                        const int32_t* row_ptr, int32_t n_groups, const float* mean, const float* rstd, void* dx, float* ws,
                        int32_t cols, int32_t relu, int32_t dtype);

/* ---- positional encoding  gnn.PositionalEncoding + add   models/graph.py:37,63 ----------
 * y[n, c] = x[n, c] + (c < C/2 ? sin(pos[n]*freq[c]) : cos(pos[n]*freq[c-C/2])) */
int egk_pe_add(egk_stream_t s, const void* x, const int64_t* pos, const float* freq, void* y, int32_t rows,
               int32_t cols, int32_t dtype);
/* The same with PE evaluated once per DISTINCT position instead of once per node (a batch of B sequences of T clips has T
 * positions): table[p - pos_min, :] = PE(p) for p in [pos_min, pos_min + n_pos) (egk_pe_table: the sinf / cosf evaluations
 * of egk_pe_add, so the sums are bit-identical), then y = x + table[pos - pos_min] (positions outside the table are
 * evaluated directly). */
int egk_pe_table(egk_stream_t s, const float* freq, int64_t pos_min, int32_t n_pos, int32_t cols, float* table);
int egk_pe_add_table(egk_stream_t s, const void* x, const int64_t* pos, const float* freq, const float* table, int64_t pos_min,
                     int32_t n_pos, void* y, int32_t rows, int32_t cols, int32_t dtype);

/* ---- CSR row gathers (message passing) ------------------------------------------------
 * out[i,:] = sum_{e in [rowptr[i], rowptr[i+1])} w_e * x[col[e], :]
 *   w_e = wgt ? wgt[e] : 1/(rowptr[i+1]-rowptr[i])      (mean; empty rows give 0)
 * if relu_gate != NULL the result row i is multiplied by (relu_gate[i,:] > 0) (fuses the
 * ReLU backward of SAGEConv's projection).  One launch = SAGEConv mean aggregation forward
 * (CSR by target, wgt NULL) or its backward (CSR by source, wgt = 1/deg(target)):
 * models/graph.py:42 (PyG SAGEConv.propagate + MeanAggregation).  No atomics:
 * bitwise reproducible.
 * heavy_rows (may be NULL with n_heavy = 0): ascending ids of EXACTLY the rows with more than
 * egk_csr_heavy_threshold() edges, listed by whoever built the CSR; those rows are summed by several workgroups
 * each (edge ranges -> f32 partial rows in ws, egk_csr_heavy_ws_bytes(n_heavy, cols) bytes, -> ordered finish)
 * instead of by one: the LTA fan-out node has out-degree T - 1.
 * heavy_mode: 0 = that (two extra launches; for rows of hundreds of edges); 1 = every listed row is summed by ONE
 * workgroup of the same launch, edges in order, while the other workgroups do the light rows (for listed rows of a few
 * dozen edges, e.g. T = 32: no extra launch, ws may be NULL; the caller knows the degrees it listed). */
int egk_csr_gather(egk_stream_t s, const void* x, const int32_t* rowptr, const int32_t* col, const float* wgt,
                   const void* relu_gate, void* out, int32_t rows, int32_t cols, int32_t dtype, const int32_t* heavy_rows,
                   int32_t n_heavy, float* ws, int32_t heavy_mode);
/* The forward (unweighted mean) orientation with the BANDED fast path: band[i] names row i's in-neighbours when they are a
 * subset of {i - 1, i, i + 1} -- bit 0: i - 1, bit 1: i, bit 2: i + 1, CSR entries of the row in ascending order -- and is
 * 0xFF for any other row (the LTA forecast nodes), which goes through rowptr / col as in egk_csr_gather.  A radius-1
 * temporal graph (RadiusGraph(r = 1.5) per sequence, reference main_temporal.py:168) is banded everywhere: its rows need no
 * index fetch at all.  Same sums in the same order as egk_csr_gather on the same CSR: bit-identical. */
int egk_csr_gather_banded(egk_stream_t s, const void* x, const int32_t* rowptr, const int32_t* col, const uint8_t* band,
                          void* out, int32_t rows, int32_t cols, int32_t dtype, const int32_t* heavy_rows, int32_t n_heavy,
                          float* ws, int32_t heavy_mode);
int64_t egk_csr_heavy_ws_bytes(int32_t n_heavy, int32_t cols);
int32_t egk_csr_heavy_threshold(void);

/* GraphONE SAGEConv(aggr='max') over cat([bank, f]) restricted to the N feature rows that are
 * kept (graphONE.py:104-115): m[n,:] = max(f[n,:], bank[nn[n,0..k),:]); arg[n,c] = winner
 * (0..k-1 = neighbour slot, k = self).  bwd: df[n,c] = (arg==k) ? dm[n,c] : 0  (bank frozen). */
int egk_gather_max_fwd(egk_stream_t s, const void* f, const float* bank, const int64_t* nn, void* m, uint8_t* arg,
                       int32_t rows, int32_t cols, int32_t k, int32_t dtype);
int egk_gather_max_bwd(egk_stream_t s, const void* dm, const uint8_t* arg, void* df, int32_t rows, int32_t cols,
                       int32_t k, int32_t accumulate, int32_t dtype);
/* egk_gather_max_fwd for the rows of 1 .. 4 GROUPS in one launch (GraphONE.interact's auxiliary tasks, graphONE.py:76-85: each
 * task has its own bank and neighbour lists): group g owns rows [g * rows, (g + 1) * rows) of f / m / arg.  Same values and
 * winners as one egk_gather_max_fwd call per group. */
int egk_gather_max_group_fwd(egk_stream_t s, const void* f, const float* const* banks, const int64_t* const* nns,
                             int32_t n_groups, void* m, uint8_t* arg, int32_t rows, int32_t cols, int32_t k, int32_t dtype);
/* Development knob: 1 (default) = rows whose loads are all requested up front (k = 4 / 8, cols a multiple of 256), 0 = the generic
 * kernel everywhere (bit-identical); < 0 only reads.  Returns the previous setting. */
int egk_gather_max_tune(int32_t up_front);

/* global_max_pool over contiguous sequences (oscc.py:68,85): out[b,:] = max_{n in [ptr[b],ptr[b+1])} x[n,:];
 * arg[b,c] = winning row.  bwd scatters dout to the winners (dx zero elsewhere). */
int egk_segment_max_fwd(egk_stream_t s, const void* x, const int32_t* ptr, void* out, int32_t* arg, int32_t n_seg,
                        int32_t cols, int32_t dtype);
int egk_segment_max_bwd(egk_stream_t s, const void* dout, const int32_t* arg, const int32_t* ptr, void* dx,
                        int32_t n_seg, int32_t rows, int32_t cols, int32_t dtype);
/* The same pools over n_src (<= 4) inputs of one shape and one sequence layout as ONE launch each way (oscc.py:68,85: the OSCC head
 * pools its own features and one GraphONE output per auxiliary task). */
int egk_segment_max_multi_fwd(egk_stream_t s, const void* const* xs, const int32_t* ptr, void* const* outs, int32_t* const* args,
                              int32_t n_src, int32_t n_seg, int32_t cols, int32_t dtype);
int egk_segment_max_multi_bwd(egk_stream_t s, const void* const* douts, const int32_t* const* args, const int32_t* ptr,
                              void* const* dxs, int32_t n_src, int32_t n_seg, int32_t rows, int32_t cols, int32_t dtype);

/* ---- cosine k-NN   GraphONE.__compute_edges + cos_dissimilarity  graphONE.py:119-151 ----
 * inv_norm[r] = 1/||x[r,:]||  (rows of features or of the bank) */
int egk_row_inv_norm(egk_stream_t s, const void* x, float* inv_norm, int32_t rows, int32_t cols, int32_t dtype);
/* dist[n,j] = 1 - dot[n,j]*f_inv[n]*b_inv[j]  (cos_dissimilarity, graphONE.py:148-151) */
int egk_cos_dist(egk_stream_t s, const float* dot, int64_t ldd, const float* f_inv, const float* b_inv, float* dist,
                 int32_t rows, int32_t K);
/* dist[n,j] = 1 - dot[n,j]*fin[n]*bin[j]; nn[n,0..k) = indices of the k smallest distances,
 * ascending, ties -> lower index (what a stable argsort gives). k <= 16. */
int egk_topk_smallest(egk_stream_t s, const float* dot, int64_t ldd, const float* f_inv, const float* b_inv,
                      int64_t* nn, int32_t rows, int32_t K, int32_t k);

/* The same selection from ONE bf16 product (reference models/graphONE/graphONE.py:119-141,148-151; replaces the f32-grade product in
 * front of egk_topk_smallest in the bf16 compute modes): dot1[n,j] = hi(f_n) . hi(p_j), hi(x) = the bf16 nearest to x (bf16 MFMA, f32
 * accumulation).  Per row the error of cos from dot1 is bounded by E = rf + rb (+ second-order and accumulation terms) with
 * rf = ||f_n - hi(f_n)|| / ||f_n|| (computed here) and *rb_max = max_j ||p_j - hi(p_j)|| / ||p_j|| (egk_bf16_residual_ratio); every
 * prototype whose approximate distance is within 2 E of the row's k-th smallest approximate distance gets its EXACT product (f32 rows
 * f [rows, H], bank [K, H], double accumulation) and egk_topk_smallest's key 1 - dot * f_inv[n] * b_inv[j]; nn[n, 0..k) = the k nearest
 * of them, ascending, ties -> lower index.  cand[n] (optional) = how many prototypes of row n were re-evaluated.  Rows of f and bank
 * must be 16-byte aligned. k <= 16. */
int egk_topk_window(egk_stream_t s, const float* dot1, int64_t ldd, const float* f, int64_t ldf, const float* bank, int64_t ldb,
                    const float* f_inv, const float* b_inv, const float* rb_max, int64_t* nn, int32_t* cand, int32_t rows, int32_t K,
                    int32_t H, int32_t k);
/* n_groups (<= 8) such searches as ONE launch: rows [g * rows_per_group, (g + 1) * rows_per_group) of dot1 / f / f_inv / nn / cand are
 * searched in banks[g] (all [K, H], row stride ldb) with b_invs[g], rb_maxs[g] -- the auxiliary tasks of one EgoPack batch, whose
 * feature rows are consecutive blocks of one buffer (graphONE.py:94-115 runs the tasks one after the other). */
int egk_topk_window_group(egk_stream_t s, const float* dot1, int64_t ldd, const float* f, int64_t ldf, const float* const* banks,
                          int64_t ldb, const float* f_inv, const float* const* b_invs, const float* const* rb_maxs, int64_t* nn,
                          int32_t* cand, int32_t n_groups, int32_t rows_per_group, int32_t K, int32_t H, int32_t k);
/* The same search with the screen's product taken from IEEE-HALF roundings of the operands (screen_f16 = 1; egk_cast_f16,
 * egk_gemm_grouped with op_f16, egk_residual_ratio16(..., 1)): 11 significand bits instead of bf16's 8 make the proven window
 * about eight times narrower -- real prototype banks (thousands of class prototypes of trained features) put 30-90 prototypes
 * inside the bf16 window of a row, 4-12 inside the f16 one.  Same lists (the candidates' distances are exact either way). */
int egk_topk_window_group16(egk_stream_t s, const float* dot1, int64_t ldd, const float* f, int64_t ldf, const float* const* banks,
                            int64_t ldb, const float* f_inv, const float* const* b_invs, const float* const* rb_maxs, int64_t* nn,
                            int32_t* cand, int32_t n_groups, int32_t rows_per_group, int32_t K, int32_t H, int32_t k, int32_t screen_f16);
int egk_residual_ratio16(egk_stream_t s, const float* x, int64_t ld, float* r, float* rmax, int32_t rows, int32_t cols, int32_t f16);
/* y[i] = the IEEE half nearest to x[i] (round to nearest even; inf beyond 65504), stored as 16-bit words */
int egk_cast_f16(egk_stream_t s, const float* x, void* y, int64_t n);
/* The grouped search's three passes over its contiguous f32 feature rows as ONE launch: inv_norm as egk_row_inv_norm (same bits),
 * hi = the bf16 rounding (what egk_cast writes), h16 (may be NULL) = the IEEE-half rounding (what egk_cast_f16 writes).
 * cols % 4 == 0.  Replaces three launches in front of the screen's product (graphONE.py:119-141). */
int egk_row_inv_norm_cast(egk_stream_t s, const float* x, float* inv_norm, void* hi, void* h16, int32_t rows, int32_t cols);
/* r[j] = ||x_j - hi(x_j)|| / ||x_j|| for the rows of an f32 matrix, *rmax = max_j r[j] (once per prototype bank) */
int egk_bf16_residual_ratio(egk_stream_t s, const float* x, int64_t ld, float* r, float* rmax, int32_t rows, int32_t cols);

/* distance_func='l2' (graphONE.py:126-127,144-145: cdist / 4096): sq_norm[r] = ||x[r,:]||^2 and
 * dist[n,j] = sqrt(max(f_sq[n] + b_sq[j] - 2 dot[n,j], 0)) / 4096; selection exactly as egk_topk_smallest. */
int egk_row_sq_norm(egk_stream_t s, const void* x, float* sq_norm, int32_t rows, int32_t cols, int32_t dtype);
int egk_topk_smallest_l2(egk_stream_t s, const float* dot, int64_t ldd, const float* f_sq, const float* b_sq,
                         int64_t* nn, int32_t rows, int32_t K, int32_t k);

/* Trainable prototypes (GraphONE(freeze=False), graphONE.py:47-49): gradient of the bank rows through the max
 * aggregation.  dbank[p,:] += sum over edges e of prototype p (t_rowptr [K+1]; t_edge[e] = n*k + j with nn[n,j] == p,
 * ascending inside a prototype) of (arg[n,:] == j ? dm[n,:] : 0).  One wave per prototype, fixed order, no atomics. */
int egk_gather_max_bank_grad(egk_stream_t s, const void* dm, const uint8_t* arg, const int32_t* t_rowptr,
                             const int32_t* t_edge, float* dbank, int32_t K, int32_t cols, int32_t k, int32_t dtype);

/* ---- prototype bank accumulation  graphone.py:53,55 (scatter(..., reduce='sum') added to a float64 bank
 * + bincount).  The labelled rows come grouped by label: order[seg_ptr[g] .. seg_ptr[g+1]) are the node ids of
 * group g in ascending node order, seg_label[g] its label (distinct per call; groups with a label outside
 * [0, n_labels) are skipped).  bank[seg_label[g],:] += (double)(fp32 sum of x[order[..],:] in node order),
 * count[seg_label[g]] += group size.  No atomics: bitwise reproducible. */
int egk_segment_sum_rows_f64(egk_stream_t s, const void* x, const int32_t* order, const int32_t* seg_ptr,
                             const int64_t* seg_label, double* bank, int64_t* count, int32_t n_seg, int32_t cols,
                             int64_t n_labels, int32_t dtype);

/* ---- losses ---------------------------------------------------------------------------
 * nn.CrossEntropyLoss(reduction='none', ignore_index=-1[, label_smoothing]) criterion/wrapper.py:67-82,
 * recognition.py:63, oscc.py:90.  loss[n] (+)= CE(logits[n,:], y[n*y_stride]); lse[n] saved.
 * bwd: dlogits[n,c] = gscale[n] * (softmax - target_dist)   (0 for ignored rows). */
int egk_ce_fwd(egk_stream_t s, const float* logits, int64_t ld, const int64_t* y, int64_t y_stride, float* loss,
               float* lse, int32_t rows, int32_t C, float smoothing, int32_t accumulate);
int egk_ce_bwd(egk_stream_t s, const float* logits, int64_t ld, const int64_t* y, int64_t y_stride, const float* lse,
               const float* gloss, void* dlogits, int64_t ldd, int32_t rows, int32_t C, float smoothing, int32_t dtype);
/* Fused multi-head form for training steps that know d objective / d loss[n] = gscale when the loss is computed (the step
 * objective is sum_t w_t * mean(loss_t), main_temporal.py:99-128): loss[n] = sum_h CE_h(n) and
 * dlogits[n, dcol[h] + c] = gscale * (softmax_h - target_h) for c < C[h], 0 for C[h] <= c < pad[h] (the zero-padded column
 * blocks of the classifier bank's gradient operand), in ONE launch.  y: [rows, y_stride] int64, head h reads column h. */
int egk_ce_fused(egk_stream_t s, const float* const* logits, const int64_t* ld, const int32_t* C, const int32_t* pad,
                 const int64_t* dcol, int32_t n_heads, const int64_t* y, int64_t y_stride, float* loss, void* dlogits, int64_t ldd,
                 int32_t rows, float smoothing, float gscale, int32_t dtype);
/* The same for up to 4 TASKS in one launch (the AR and LTA heads of a multi-task step, main_temporal.py:93-126: one cross
 * entropy per task over its own logits, labels, loss vector and gradient operand).  Per task the arithmetic of
 * egk_ce_fused: bit-identical results. */
typedef struct {
    const float* logits[4];
    int64_t ld[4];
    int32_t C[4];
    int32_t pad[4];
    int64_t dcol[4];
    int32_t n_heads;
    const int64_t* y;
    int64_t y_stride;
    float* loss;
    void* dlogits;
    int64_t ldd;
    int32_t rows;
    float gscale;
} egk_ce_task;
int egk_ce_fused_multi(egk_stream_t s, const egk_ce_task* tasks, int32_t count, float smoothing, int32_t dtype);
/* nn.BCEWithLogitsLoss(reduction='none') on y.float()  main_temporal.py:123,298; pnr.py:82-83 */
int egk_bce_fwd(egk_stream_t s, const float* logits, const int64_t* y, float* loss, int32_t n);
int egk_bce_bwd(egk_stream_t s, const float* logits, const int64_t* y, const float* gloss, void* dlogits, int32_t n,
                int32_t dtype);

/* OSCCTask.compute_loss 'bce' / 'focal' (models/tasks/oscc.py:91-96): elementwise sigmoid losses of [rows, C] logits against
 * one_hot(y, C).float(), reduction 'none'.  kind 0 = F.binary_cross_entropy_with_logits; kind 1 =
 * torchvision.ops.sigmoid_focal_loss(alpha, gamma) (alpha < 0: unweighted).  y must hold class ids in [0, C). */
int egk_onehot_sigmoid_loss_fwd(egk_stream_t s, const float* logits, const int64_t* y, float* loss, int32_t rows, int32_t C,
                                int32_t kind, float alpha, float gamma);
int egk_onehot_sigmoid_loss_bwd(egk_stream_t s, const float* logits, const int64_t* y, const float* gloss, void* dlogits,
                                int32_t rows, int32_t C, int32_t kind, float alpha, float gamma, int32_t dtype);

/* Activation element types.  Every entry point with a trailing ``dtype`` argument reads / writes its
 * [rows, cols] activation (and activation-gradient) matrices as EGK_F32 or EGK_BF16; all arithmetic,
 * statistics, parameters and parameter gradients stay f32.
 * egk_cast converts n contiguous elements between the two types. */
int egk_cast(egk_stream_t s, const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n);
/* row-strided variant: dst[r, c] = src[r, c], c < cols, with independent leading dimensions (elements) and
 * element types (equal types allowed: a re-striding copy).  Used to give a [rows, cols] gradient whose width is
 * not a multiple of 8 a 16-byte aligned row stride before it becomes a contraction operand; columns
 * [cols, zero_cols) of every destination row are set to zero (zero_cols <= ld_dst; 0 for none), which makes the
 * copy a valid contraction operand over a K axis padded to the 64-deep tile of the pipelined kernel. */
int egk_cast_rows(egk_stream_t s, const void* src, int32_t src_dtype, int64_t ld_src, void* dst, int32_t dst_dtype,
                  int64_t ld_dst, int32_t rows, int32_t cols, int32_t zero_cols);

/* ---- small elementwise helpers ----------------------------------------------------------- */
/* y = keep ? x/(1-p) : 0 with a fresh Philox mask (nn.Dropout: task.py:18, graph.py:30, heads) */
int egk_dropout_fwd(egk_stream_t s, const void* x, void* y, uint8_t* mask, int64_t n, float p, uint64_t seed,
                    uint64_t offset, const uint64_t* dev_offset, int32_t dtype);
int egk_dropout_bwd(egk_stream_t s, const void* dy, const uint8_t* mask, void* dx, int64_t n, float p, int32_t dtype);
/* dx = y > 0 ? dy : 0  -- backward of a ReLU fused into a contraction epilogue (graph.py:42 project) */
int egk_relu_gate(egk_stream_t s, const void* dy, const void* y, void* dx, int64_t n, int32_t dtype);
/* out = a*x + b*y (y may be NULL) */
int egk_axpby(egk_stream_t s, const float* x, const float* y, float* out, int64_t n, float a, float b);
/* out[i] = scalar[0] * coef  -- backward of the scaled mean below */
int egk_fill_scaled(egk_stream_t s, const float* scalar, float coef, float* out, int64_t n);
/* out[0] (+)= scale * sum(x[0..n))  -- loss.mean() * weight, deterministic single-block tree */
int egk_sum_scale(egk_stream_t s, const float* x, float* out, int64_t n, float scale, int32_t accumulate);

/* p[0 .. bytes) = 0 (16-byte aligned, whole 16-byte groups): ``optimizer.zero_grad()`` of the flat gradient buffer
 * (reference main_temporal.py:77) as a launch of the library */
int egk_zero_fill(egk_stream_t s, void* p, int64_t bytes);
/* n_ranges (1 .. 48) byte ranges [begin[i], begin[i] + bytes[i]) of ``base`` cleared by ONE launch (whole 16-byte groups; HOST
 * arrays): the gradient slots that are still ACCUMULATED into, once the weight matrices whose single gradient launch stores its
 * result (accumulate = 0) need no clear (egopack_amd.optim.FlatAdam.store_slots; the reference's optimizer.zero_grad(),
 * main_temporal.py:76). */
int egk_zero_fill_ranges(egk_stream_t s, void* base, const int64_t* begin, const int64_t* bytes, int32_t n_ranges);

/* ---- optimiser  torch.optim.Adam (L2 weight decay)  configs/defaults.yaml:17-20 ----------
 * One launch over the flat parameter / gradient / moment buffers.  hyper (device, float[4]) =
 * {lr, 1-beta1^t, sqrt(1-beta2^t), grad_scale}: rewritten by the host between graph replays.
 * g' = g*grad_scale + wd*p; m = b1*m+(1-b1)*g'; v = b2*v+(1-b2)*g'^2;
 * p -= (lr/bc1) * m / (sqrt(v)/bc2_sqrt + eps)
 * g_dtype: EGK_F32, or EGK_BF16 when the gradient buffer handed in is the bf16 copy a compressed all-reduce summed.
 * bf16_shadow (may be NULL): bf16 copy of the updated parameters, same flat layout, written by the same
 * launch -- the operand the bf16 contractions read (no separate cast pass over the weights). */
int egk_adam_step(egk_stream_t s, float* p, const void* g, int32_t g_dtype, float* m, float* v, int64_t n,
                  const float* hyper, float beta1, float beta2, float eps, float weight_decay, void* bf16_shadow);
/* the same launch with two riders: bf16_lo_shadow (or NULL) receives bf16(p - bf16(p)) of the updated parameters -- the LOW halves
 * of the weight operands of the three-product (f32-grade) contractions, egk_split_bf16's bits, so that the forward-only precise
 * pass of the next step needs no split launch over the weights; and *bump_word += bump (bump_word or NULL) by one thread of the
 * launch (n > 0): a device-side per-step counter -- the Philox offset word the step's dropout launches add to their offsets --
 * moves on inside the optimizer's launch instead of in a launch of its own */
int egk_adam_step_bump(egk_stream_t s, float* p, const void* g, int32_t g_dtype, float* m, float* v, int64_t n,
                       const float* hyper, float beta1, float beta2, float eps, float weight_decay, void* bf16_shadow,
                       void* bf16_lo_shadow, int64_t* bump_word, int64_t bump);
/* The constants of the NEXT step computed on the device: t = ++(*t_dev) (device int64: optimizer steps taken so far);
 * hyper[4] = {src[0] = lr, 1 - beta1^t, sqrt(1 - beta2^t), src[1] = grad_scale} (double pow / sqrt, rounded to f32 once, as
 * torch.optim.Adam's bias corrections are).  One thread; a node of the captured step, so that a graph replay needs no
 * host -> device copy in front of it and the step count advances with the replays.  src: device float[2]. */
int egk_adam_hyper(egk_stream_t s, const float* src, int64_t* t_dev, double beta1, double beta2, float* hyper);

/* ---- the step objective in one launch each way  main_temporal.py:99-128 (torch.stack([w * l.mean() ...]).sum()) ----
 * out[0] = sum_k coefs[k] * sum(xs[k][0..ns[k])), terms added in k order (count <= 8; xs / ns / coefs are HOST arrays);
 * backward: outs[k][i] = scalar[0] * coefs[k]. */
int egk_weighted_sums(egk_stream_t s, const float* const* xs, const int64_t* ns, const float* coefs, int32_t count, float* out);
/* ... and acc[k] += sum(xs[k]) (device double[count], may be NULL): the per-task loss sums a training loop logs per epoch
 * (main_temporal.py:129-134 keeps them with .item() per step) accumulate inside the step instead of in launches between steps. */
int egk_weighted_sums_acc(egk_stream_t s, const float* const* xs, const int64_t* ns, const float* coefs, int32_t count, float* out,
                          double* acc);
int egk_fill_scaled_multi(egk_stream_t s, const float* scalar, const float* coefs, float* const* outs, const int64_t* ns,
                          int32_t count);
/* dst = srcs[0] | srcs[1] | ... : count <= 8 contiguous blocks of nbytes[k] bytes each, a NULL source zero-fills its
 * block (the per-task feature gradients back into the merged buffer of the fused backbone pass); host arrays. */
int egk_copy_blocks(egk_stream_t s, const void* const* srcs, const int64_t* nbytes, void* dst, int32_t count);

/* ---- resident feature store  data/base_dataset.py:128-155, ego4d_fho.py:217-242 ----
 * out[i, :] = table[idx[i], :] for i < n, zeros where idx[i] < 0 or >= table_rows (the reference's all-zero clip when
 * a window cannot be sampled); table rows are ld elements apart; element types EGK_F32 / EGK_BF16 independently
 * (bf16 -> bf16 and f32 -> f32 copy bits; f32 -> bf16 rounds to nearest even).  The device-side half of
 * ``np.take(video_features[a:b], indices, axis=0)``: the index arithmetic stays on the host, bit-exact. */
int egk_gather_rows(egk_stream_t s, const void* table, int32_t table_dtype, int64_t ld, int64_t table_rows,
                    const int64_t* idx, void* out, int32_t out_dtype, int64_t n, int32_t cols);

/* interpolating variant (PNR key-frame sampling data/ego4d_oscc.py:258-275, LTA 'avg' forecast nodes ego4d_fho.py:388-394):
 * out[i, :] = table[lo[i], :] where lo[i] == hi[i], else (float)((1 - w[i]) * table[lo[i], :] + w[i] * table[hi[i], :]) in
 * double with separately rounded products and sum (numpy's evaluation, then ``.float()``); an index < 0 or >= table_rows
 * stands for an all-zero row. */
int egk_gather_lerp_rows(egk_stream_t s, const void* table, int32_t table_dtype, int64_t ld, int64_t table_rows,
                         const int64_t* lo, const int64_t* hi, const double* w, void* out, int32_t out_dtype, int64_t n,
                         int32_t cols);

/* ---- validation metrics  utils/meters/ego4d.py, utils/meters/utils.py:6-28 ----
 * rank[r] = #{j : s[r,j] > s[r,y]} + #{j < y : s[r,j] == s[r,y]} for y = labels[r*label_stride]; -1 when y < 0
 * (ignore_index) or y >= C.  top-k accuracy = mean(rank < k) over rank >= 0; per-class recall likewise. */
int egk_label_rank(egk_stream_t s, const float* logits, int64_t ld, const int64_t* labels, int64_t label_stride,
                   int32_t* rank, int32_t rows, int32_t C);
/* out[n*K + k] = Levenshtein distance between pred[n, :, k] and label[n, :] (element strides given), Z <= 64
 * (Ego4dLTAMeter._edit_distance, ego4d.py:410-423: editdistance.eval(...)/Z, minimum over k taken by the caller) */
int egk_edit_distance(egk_stream_t s, const int64_t* pred, int64_t p_sn, int64_t p_sz, int64_t p_sk, const int64_t* label,
                      int64_t l_sn, int64_t l_sz, int32_t* out, int32_t N, int32_t Z, int32_t K);

#ifdef __cplusplus
}
#endif
#endif /* EGOPACK_HIP_H */
