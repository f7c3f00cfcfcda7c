// Loss kernels (cross-entropy with ignore_index / label smoothing, BCE-with-logits), small
// elementwise helpers (dropout, axpby, scaled sum) and the single-launch Adam step over the flat
// parameter buffer.  All HBM-bound / latency-bound.
#include <math.h>

#include "common.h"

namespace egk {

constexpr int WPB = 4;

// ---- cross entropy: one wave per row ------------------------------------------------------------
// loss = lse - (1-eps)*x_y - eps/C * sum_c x_c     (0 for y == -1)
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long long ld,
                                                     const long long* __restrict__ y, long long ys, float* __restrict__ loss,
                                                     float* __restrict__ lse, int rows, int C, float smoothing,
                                                     int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const float* lr = logits + (long long)row * ld;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lr[c]);
        mx = wave_max(mx);
        float se = 0.f, sx = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = lr[c];
            se += expf(v - mx);
            sx += v;
        }
        se = wave_sum(se);
        sx = wave_sum(sx);
        if (lane == 0) {
            const float l = mx + logf(se);
            lse[row] = l;
            const long long t = y[(long long)row * ys];
            float o = 0.f;
            if (t >= 0 && t < C) o = l - (1.f - smoothing) * lr[t] - (smoothing > 0.f ? smoothing / C * sx : 0.f);
            loss[row] = accumulate ? loss[row] + o : o;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long long ld,
                                                     const long long* __restrict__ y, long long ys,
                                                     const float* __restrict__ lse, const float* __restrict__ gloss,
                                                     T* __restrict__ dlogits, long long ldd, int rows, int C,
                                                     float smoothing) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        const float* lr = logits + (long long)row * ld;
        T* dr = dlogits + (long long)row * ldd;
        const long long t = y[(long long)row * ys];
        const bool live = t >= 0 && t < C;
        const float g = live ? gloss[row] : 0.f;
        const float l = lse[row];
        const float sm = smoothing > 0.f ? smoothing / C : 0.f;
        for (int c = lane; c < C; c += 64) {
            float d = 0.f;
            if (live) d = g * (expf(lr[c] - l) - (c == t ? 1.f - smoothing : 0.f) - sm);
            st1t(dr + c, d);
        }
    }
}

// ---- fused multi-head cross entropy: loss AND its gradient in one launch -----------------------------------------------------
// The training heads of the engine know the gradient of the objective with respect to every loss-vector element when the
// loss is computed: objective = sum_t w_t * mean(loss_t)  =>  d objective / d loss_t[n] = w_t / N_t, a constant.  One wave
// per row then does, for every head h of the task (verb, noun): loss[n] += CE_h(n) and
//   dlogits_h[n, c] = gscale * (softmax_h(n)[c] - target_h(n)[c])      (0 for ignored rows)
// written straight into the classifier bank's operand buffer, zero-filling the bank's pad columns [C_h, pad_h) on the way
// (the buffer needs no memset).  Replaces 2 forward + 2 backward launches per two-head task.
constexpr int CE_MAX_HEADS = 4;
struct CEHeads {
    const float* logits[CE_MAX_HEADS];
    long long ld[CE_MAX_HEADS];
    int C[CE_MAX_HEADS];
    int pad[CE_MAX_HEADS];     // columns [C, pad) of the gradient block are set to zero
    long long dcol[CE_MAX_HEADS];  // first column of the head's block in the gradient buffer
};
template <typename T>
__global__ __launch_bounds__(256) void ce_fused_kernel(const CEHeads H, int n_heads, const long long* __restrict__ y, long long ys,
                                                       float* __restrict__ loss, T* __restrict__ dlogits, long long ldd, int rows,
                                                       float smoothing, float gscale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        float total = 0.f;
        for (int h = 0; h < n_heads; ++h) {
            const float* lr = H.logits[h] + (long long)row * H.ld[h];
            const int C = H.C[h];
            T* dr = dlogits + (long long)row * ldd + H.dcol[h];
            const long long t = y[(long long)row * ys + h];
            const bool live = t >= 0 && t < C;
            float mx = -INFINITY;
            for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lr[c]);
            mx = wave_max(mx);
            float se = 0.f, sx = 0.f;
            for (int c = lane; c < C; c += 64) {
                const float v = lr[c];
                se += expf(v - mx);
                sx += v;
            }
            se = wave_sum(se);
            sx = wave_sum(sx);
            const float l = mx + logf(se);
            if (live) total += l - (1.f - smoothing) * lr[t] - (smoothing > 0.f ? smoothing / C * sx : 0.f);
            const float sm = smoothing > 0.f ? smoothing / C : 0.f;
            for (int c = lane; c < H.pad[h]; c += 64) {
                float d = 0.f;
                if (live && c < C) d = gscale * (expf(lr[c] - l) - (c == t ? 1.f - smoothing : 0.f) - sm);
                st1t(dr + c, d);
            }
        }
        if (lane == 0) loss[row] = total;
    }
}

// the same for several TASKS in one launch (blockIdx.y = task): the AR and LTA heads of a multi-task step
constexpr int CE_MAX_TASKS = 4;
struct CETasks {
    CEHeads H[CE_MAX_TASKS];
    int n_heads[CE_MAX_TASKS];
    const long long* y[CE_MAX_TASKS];
    long long ys[CE_MAX_TASKS];
    float* loss[CE_MAX_TASKS];
    void* dlogits[CE_MAX_TASKS];
    long long ldd[CE_MAX_TASKS];
    int rows[CE_MAX_TASKS];
    float gscale[CE_MAX_TASKS];
};
template <typename T>
__global__ __launch_bounds__(256) void ce_fused_multi_kernel(const CETasks P, float smoothing) {
    const int k = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const CEHeads& H = P.H[k];
    const int n_heads = P.n_heads[k], rows = P.rows[k];
    const long long* __restrict__ y = P.y[k];
    const long long ys = P.ys[k], ldd = P.ldd[k];
    float* __restrict__ loss = P.loss[k];
    T* __restrict__ dlogits = (T*)P.dlogits[k];
    const float gscale = P.gscale[k];
    const RowWalk rw = row_walk(blockIdx.x, gridDim.x, 0, rows, wave, WPB);  // (XCD x owns a contiguous eighth of the rows: common.h)
    for (int row = rw.first; row < rw.end; row += rw.step) {
        float total = 0.f;
        for (int h = 0; h < n_heads; ++h) {  // (the arithmetic of ce_fused_kernel, statement for statement)
            const float* lr = H.logits[h] + (long long)row * H.ld[h];
            const int C = H.C[h];
            T* dr = dlogits + (long long)row * ldd + H.dcol[h];
            const long long t = y[(long long)row * ys + h];
            const bool live = t >= 0 && t < C;
            float mx = -INFINITY;
            for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lr[c]);
            mx = wave_max(mx);
            float se = 0.f, sx = 0.f;
            for (int c = lane; c < C; c += 64) {
                const float v = lr[c];
                se += expf(v - mx);
                sx += v;
            }
            se = wave_sum(se);
            sx = wave_sum(sx);
            const float l = mx + logf(se);
            if (live) total += l - (1.f - smoothing) * lr[t] - (smoothing > 0.f ? smoothing / C * sx : 0.f);
            const float sm = smoothing > 0.f ? smoothing / C : 0.f;
            for (int c = lane; c < H.pad[h]; c += 64) {
                float d = 0.f;
                if (live && c < C) d = gscale * (expf(lr[c] - l) - (c == t ? 1.f - smoothing : 0.f) - sm);
                st1t(dr + c, d);
            }
        }
        if (lane == 0) loss[row] = total;
    }
}

// ---- BCE with logits ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ y,
                                                      float* __restrict__ loss, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i], t = (float)y[i];
    // torch: (1 - t) * x + max(-x, 0) + log1p(exp(-|x|))
    loss[i] = (1.f - t) * v + fmaxf(-v, 0.f) + log1pf(expf(-fabsf(v)));
}
template <typename T>
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ y,
                                                      const float* __restrict__ gloss, T* __restrict__ dx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    st1t(dx + i, (1.f / (1.f + expf(-v)) - (float)y[i]) * gloss[i]);
}

// ---- sigmoid losses against one-hot class targets (OSCCTask.compute_loss 'bce' / 'focal', reference oscc.py:91-96) ---
// element i = (row, c) of [rows, C] logits; target t = (y[row] == c).  kind 0: BCE-with-logits; kind 1: torchvision
// sigmoid_focal_loss(alpha, gamma):  p_t = sigmoid(z), z = (2t-1) x;  L = a_t (1-p_t)^gamma (-log p_t),
// a_t = alpha t + (1-alpha)(1-t) (alpha < 0: no weighting);  dL/dx = (2t-1) a_t (1-p_t)^gamma (gamma p_t log p_t - (1-p_t)).
__device__ __forceinline__ float log_sigmoid(float z) { return -(fmaxf(-z, 0.f) + log1pf(expf(-fabsf(z)))); }

__global__ __launch_bounds__(256) void onehot_sigmoid_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ y,
                                                                 float* __restrict__ loss, int n, int C, int kind, float alpha,
                                                                 float gamma) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float t = y[i / C] == (long long)(i % C) ? 1.f : 0.f;
    const float ce = (1.f - t) * v + fmaxf(-v, 0.f) + log1pf(expf(-fabsf(v)));
    if (kind == 0) {
        loss[i] = ce;
        return;
    }
    const float p = 1.f / (1.f + expf(-v));
    const float pt = p * t + (1.f - p) * (1.f - t);
    float l = ce * (gamma == 2.f ? (1.f - pt) * (1.f - pt) : powf(1.f - pt, gamma));
    if (alpha >= 0.f) l *= alpha * t + (1.f - alpha) * (1.f - t);
    loss[i] = l;
}
template <typename T>
__global__ __launch_bounds__(256) void onehot_sigmoid_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ y,
                                                                 const float* __restrict__ gloss, T* __restrict__ dx, int n,
                                                                 int C, int kind, float alpha, float gamma) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float t = y[i / C] == (long long)(i % C) ? 1.f : 0.f;
    float d;
    if (kind == 0) {
        d = 1.f / (1.f + expf(-v)) - t;
    } else {
        const float sg = 2.f * t - 1.f, z = sg * v;
        const float pt = 1.f / (1.f + expf(-z)), q = 1.f - pt;
        const float mod = gamma == 2.f ? q * q : powf(q, gamma);
        const float a = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
        d = sg * a * mod * (gamma * pt * log_sigmoid(z) - q);
    }
    st1t(dx + i, d * gloss[i]);
}

// ---- dropout ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ mask, long long n, float p, uint64_t seed,
                                                          uint64_t offset, const uint64_t* __restrict__ dev_offset) {
    if (dev_offset) offset += dev_offset[0];
    const float inv = 1.f / (1.f - p);
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += (long long)gridDim.x * blockDim.x) {
        const uint4 r = philox4x32_10(offset + (uint64_t)q, seed);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
        for (int t = 0; t < 4; ++t) {
            const long long i = q * 4 + t;
            if (i < n) {
                const bool keep = u01(rr[t]) >= p;
                mask[i] = keep;
                st1t(y + i, keep ? ld1t(x + i) * inv : 0.f);
            }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                          T* __restrict__ dx, long long n, float p) {
    const float inv = 1.f / (1.f - p);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        st1t(dx + i, mask[i] ? ld1t(dy + i) * inv : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void relu_gate_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                        T* __restrict__ dx, long long n, int vec) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
         i += (long long)gridDim.x * blockDim.x * 4) {
        if (vec && i + 4 <= n) {
            const float4 g = ld4t(dy + i, 0, 4, true);
            const float4 v = ld4t(y + i, 0, 4, true);
            st4t(dx + i, 0, 4, true,
                 make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f));
        } else
            for (long long j = i; j < n && j < i + 4; ++j) st1t(dx + j, ld1t(y + j) > 0.f ? ld1t(dy + j) : 0.f);
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                    float* __restrict__ out, long long n, float a, float b, int vec) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
         i += (long long)gridDim.x * blockDim.x * 4) {
        if (vec && i + 4 <= n) {
            const float4 xv = *reinterpret_cast<const float4*>(x + i);
            float4 o = make_float4(a * xv.x, a * xv.y, a * xv.z, a * xv.w);
            if (y) {
                const float4 yv = *reinterpret_cast<const float4*>(y + i);
                o.x += b * yv.x; o.y += b * yv.y; o.z += b * yv.z; o.w += b * yv.w;
            }
            *reinterpret_cast<float4*>(out + i) = o;
        } else
            for (long long j = i; j < n && j < i + 4; ++j) out[j] = a * x[j] + (y ? b * y[j] : 0.f);
    }
}

__global__ __launch_bounds__(256) void fill_scaled_kernel(const float* __restrict__ scalar, float coef,
                                                          float* __restrict__ out, long long n) {
    const float v = scalar[0] * coef;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = v;
}

// single workgroup, fixed-order tree: bitwise reproducible loss scalars
__global__ __launch_bounds__(1024) void sum_scale_kernel(const float* __restrict__ x, float* __restrict__ out, long long n,
                                                         float scale, int accumulate) {
    __shared__ float part[16];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += part[i];
        t *= scale;
        out[0] = accumulate ? out[0] + t : t;
    }
}

// The objective of a multi-task step in ONE launch each way: out = sum_i coef_i * sum(x_i) (vectors reduced one after
// the other by the same fixed-order tree, terms added in vector order: the values of the per-vector launches), and
// its backward d x_i[:] = g * coef_i.  (Per task that was a chain of 1-workgroup launches separated by graph-node
// latencies: 7 nodes for three tasks.)
constexpr int MAXVEC = 8;
struct VecList {
    const float* x[MAXVEC];
    float* out[MAXVEC];
    long long n[MAXVEC];
    float coef[MAXVEC];
    int count;
};

__global__ __launch_bounds__(1024) void weighted_sums_kernel(const VecList v, float* __restrict__ out, double* __restrict__ acc) {
    __shared__ float part[16];
    float total = 0.f;
    for (int k = 0; k < v.count; ++k) {
        float s = 0.f;
        for (long long i = threadIdx.x; i < v.n[k]; i += 1024) s += v.x[k][i];
        s = wave_sum(s);
        __syncthreads();  // part[] free (previous vector consumed)
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int i = 0; i < 16; ++i) t += part[i];
            if (acc) acc[k] += (double)t;  // (running per-vector sums over the steps of a training loop: egk_weighted_sums_acc)
            t *= v.coef[k];
            total = k ? total + t : t;
        }
    }
    if (threadIdx.x == 0) out[0] = total;
}

__global__ __launch_bounds__(256) void fill_scaled_multi_kernel(const float* __restrict__ scalar, const VecList v) {
    const int k = blockIdx.y;
    const float val = scalar[0] * v.coef[k];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < v.n[k]; i += (long long)gridDim.x * blockDim.x)
        v.out[k][i] = val;
}

// dst = srcs[0] | srcs[1] | ... (contiguous blocks, byte sizes; a NULL source fills its block with zeros) in one launch:
// the per-task feature gradients of the fused backbone pass go back into ONE buffer (ops._SplitRows.backward).
struct BlockList {
    const unsigned char* src[MAXVEC];
    long long off[MAXVEC], bytes[MAXVEC];
    int count;
};

__global__ __launch_bounds__(256) void copy_blocks_kernel(const BlockList b, unsigned char* __restrict__ dst) {
    const int k = blockIdx.y;
    const unsigned char* s = b.src[k];
    unsigned char* d = dst + b.off[k];
    const long long n = b.bytes[k];
    const bool vec = s == nullptr ? ((uintptr_t)d & 15) == 0 : (((uintptr_t)s | (uintptr_t)d) & 15) == 0;
    const long long n16 = vec ? n / 16 : 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x)
        reinterpret_cast<uint4*>(d)[i] = s ? reinterpret_cast<const uint4*>(s)[i] : make_uint4(0, 0, 0, 0);
    for (long long i = n16 * 16 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        d[i] = s ? s[i] : 0;
}

// The step's constants ON THE DEVICE: t = ++(*t_dev); hyper = {lr, 1 - b1^t, sqrt(1 - b2^t), grad_scale} in the host's
// arithmetic (double pow / sqrt, one rounding to f32).  A node of the captured step: a replay needs no host -> device copy
// in front of it (4 us of copy + the gap behind it, per step) and the step count advances with the replays.
__global__ void adam_hyper_kernel(const float* __restrict__ src, long long* __restrict__ t_dev, double b1, double b2,
                                  float* __restrict__ hyper) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long t = *t_dev + 1;
        *t_dev = t;
        hyper[0] = src[0];
        hyper[1] = (float)(1.0 - pow(b1, (double)t));
        hyper[2] = (float)sqrt(1.0 - pow(b2, (double)t));
        hyper[3] = src[1];
    }
}

// ---- Adam (torch.optim.Adam single-tensor formulas, L2 weight decay) ----------------------------------------
// the elements [0, n) of one span, grid-stride over ``nblk`` workgroups (adam_kernel: the whole slice)
template <typename GT>
__device__ __forceinline__ void adam_span(float* __restrict__ p, const GT* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                          long long n, const AdamConsts& ac, bf16_t* __restrict__ shadow, bf16_t* __restrict__ shadow_lo,
                                          int blk, int nblk) {
    for (long long i = ((long long)blk * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)nblk * blockDim.x * 4) {
        if (i + 4 <= n) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            const float4 gv = ld4t(g + i, 0, 4, true);
            float4 mv = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
            for (int t = 0; t < 4; ++t) adam_update(pp[t], gp[t], mp[t], vp[t], ac);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
            if (shadow) st4t(shadow + i, 0, 4, true, pv);
            if (shadow_lo) {  // the LOW halves of the three-product contractions' weight operands: bf16(p - bf16(p)), egk_split_bf16's bits
                const float lo4[4] = {pp[0] - bf2f(f2bf(pp[0])), pp[1] - bf2f(f2bf(pp[1])), pp[2] - bf2f(f2bf(pp[2])), pp[3] - bf2f(f2bf(pp[3]))};
                st4t(shadow_lo + i, 0, 4, true, make_float4(lo4[0], lo4[1], lo4[2], lo4[3]));
            }
        } else {
            for (long long j = i; j < n; ++j) {
                adam_update(p[j], ld1t(g + j), m[j], v[j], ac);
                if (shadow) shadow[j] = f2bf(p[j]);
                if (shadow_lo) shadow_lo[j] = f2bf(p[j] - bf2f(f2bf(p[j])));
            }
        }
    }
}

template <typename GT>  // GT: element type of the gradient buffer (f32, or bf16 after a compressed all-reduce)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const GT* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, const float* __restrict__ hyper,
                                                   float b1, float b2, float eps, float wd, bf16_t* __restrict__ shadow,
                                                   bf16_t* __restrict__ shadow_lo, long long* __restrict__ bump_word, long long bump) {
    // (egk_adam_step_bump: a device-side counter that moves on once per step -- the Philox offset word of the step's dropout
    //  launches -- rides in this launch instead of costing one of its own)
    if (bump_word && blockIdx.x == 0 && threadIdx.x == 0) *bump_word += bump;
    const AdamConsts ac{hyper[0] / hyper[1], hyper[2], hyper[3], b1, b2, eps, wd};
    adam_span(p, g, m, v, n, ac, shadow, shadow_lo, blockIdx.x, gridDim.x);
}

static inline unsigned ew_grid(long long n, int per_thread) {
    long long b = (n / per_thread + 255) / 256;
    return (unsigned)(b < 1 ? 1 : b > 4096 ? 4096 : b);
}
static inline int row_grid(int rows) {
    int g = cdiv(rows, WPB);
    return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

}  // namespace egk

using namespace egk;

static int g_zero_fill_blocks = 0;  // development knob: egk_tune(5, workgroups); 0 = default
static int g_adam_blocks = 0;       // development knob: egk_tune(6, workgroups); 0 = default
namespace egk { void set_zero_fill_blocks(int n) { g_zero_fill_blocks = n; } }
namespace egk { void set_adam_blocks(int n) { g_adam_blocks = n; } }

extern "C" {

int egk_ce_fwd(egk_stream_t stream, const float* logits, int64_t ld, const int64_t* y, int64_t y_stride, float* loss,
               float* lse, int32_t rows, int32_t C, float smoothing, int32_t accumulate) {
    EGK_REQUIRE(logits && y && loss && lse, "egk_ce_fwd: null pointer");
    EGK_REQUIRE(C >= 1, "egk_ce_fwd: C must be >= 1");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_CE_FWD, s, 0, 4.0 * rows * C);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(row_grid(rows)), dim3(256), 0, s, logits, (long long)ld, (const long long*)y,
                       (long long)y_stride, loss, lse, rows, C, smoothing, accumulate);
    return check_launch("egk_ce_fwd");
}

int egk_ce_bwd(egk_stream_t stream, const float* logits, int64_t ld, const int64_t* y, int64_t y_stride, const float* lse,
               const float* gloss, void* dlogits, int64_t ldd, int32_t rows, int32_t C, float smoothing, int32_t dtype) {
    EGK_REQUIRE(logits && y && lse && gloss && dlogits, "egk_ce_bwd: null pointer");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_CE_BWD, s, 0, 8.0 * rows * C);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(ce_bwd_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, s, logits, (long long)ld,
                                             (const long long*)y, (long long)y_stride, lse, gloss, (T*)dlogits, (long long)ldd, rows,
                                             C, smoothing));
    return check_launch("egk_ce_bwd");
}

int egk_ce_fused(egk_stream_t stream, const float* const* logits, const int64_t* ld, const int32_t* C, const int32_t* pad,
                 const int64_t* dcol, int32_t n_heads, const int64_t* y, int64_t y_stride, float* loss, void* dlogits, int64_t ldd,
                 int32_t rows, float smoothing, float gscale, int32_t dtype) {
    EGK_REQUIRE(logits && ld && C && pad && dcol && y && loss && dlogits, "egk_ce_fused: null pointer");
    EGK_REQUIRE(n_heads >= 1 && n_heads <= CE_MAX_HEADS, "egk_ce_fused: 1 .. %d heads", CE_MAX_HEADS);
    if (rows == 0) return 0;
    CEHeads H;
    double bytes = 0;
    for (int h = 0; h < CE_MAX_HEADS; ++h) {
        const int k = h < n_heads ? h : n_heads - 1;
        EGK_REQUIRE(logits[k] && C[k] >= 1 && pad[k] >= C[k], "egk_ce_fused: bad head %d", k);
        H.logits[h] = logits[k]; H.ld[h] = ld[k]; H.C[h] = C[k]; H.pad[h] = pad[k]; H.dcol[h] = dcol[k];
        if (h < n_heads) bytes += 6.0 * rows * C[k];
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_CE_FWD, s, 0, bytes);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(ce_fused_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, s, H, n_heads, (const long long*)y,
                                             (long long)y_stride, loss, (T*)dlogits, (long long)ldd, rows, smoothing, gscale));
    return check_launch("egk_ce_fused");
}

int egk_ce_fused_multi(egk_stream_t stream, const egk_ce_task* tasks, int32_t count, float smoothing, int32_t dtype) {
    EGK_REQUIRE(tasks && count >= 1 && count <= CE_MAX_TASKS, "egk_ce_fused_multi: 1 .. %d tasks", CE_MAX_TASKS);
    CETasks P;
    double bytes = 0;
    int max_rows = 0;
    for (int i = 0; i < CE_MAX_TASKS; ++i) {
        const egk_ce_task& t = tasks[i < count ? i : count - 1];
        EGK_REQUIRE(t.y && t.loss && t.dlogits && t.n_heads >= 1 && t.n_heads <= CE_MAX_HEADS && t.rows >= 0,
                    "egk_ce_fused_multi: bad task %d", i);
        for (int h = 0; h < CE_MAX_HEADS; ++h) {
            const int k = h < t.n_heads ? h : t.n_heads - 1;
            EGK_REQUIRE(t.logits[k] && t.C[k] >= 1 && t.pad[k] >= t.C[k], "egk_ce_fused_multi: bad head %d of task %d", k, i);
            P.H[i].logits[h] = t.logits[k]; P.H[i].ld[h] = t.ld[k]; P.H[i].C[h] = t.C[k]; P.H[i].pad[h] = t.pad[k];
            P.H[i].dcol[h] = t.dcol[k];
            if (i < count && h < t.n_heads) bytes += 6.0 * t.rows * t.C[k];
        }
        P.n_heads[i] = t.n_heads; P.y[i] = (const long long*)t.y; P.ys[i] = t.y_stride; P.loss[i] = t.loss;
        P.dlogits[i] = t.dlogits; P.ldd[i] = t.ldd; P.rows[i] = i < count ? t.rows : 0; P.gscale[i] = t.gscale;
        if (i < count && t.rows > max_rows) max_rows = t.rows;
    }
    if (max_rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_CE_FWD, s, 0, bytes);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(ce_fused_multi_kernel<T>, dim3(row_grid(max_rows), count), dim3(256), 0, s, P, smoothing));
    return check_launch("egk_ce_fused_multi");
}

int egk_bce_fwd(egk_stream_t stream, const float* logits, const int64_t* y, float* loss, int32_t n) {
    EGK_REQUIRE(logits && y && loss, "egk_bce_fwd: null pointer");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_BCE_FWD, s, 0, 16.0 * n);
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, logits, (const long long*)y, loss, n);
    return check_launch("egk_bce_fwd");
}

int egk_bce_bwd(egk_stream_t stream, const float* logits, const int64_t* y, const float* gloss, void* dlogits, int32_t n,
                int32_t dtype) {
    EGK_REQUIRE(logits && y && gloss && dlogits, "egk_bce_bwd: null pointer");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_BCE_BWD, s, 0, 20.0 * n);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(bce_bwd_kernel<T>, dim3(cdiv(n, 256)), dim3(256), 0, s, logits, (const long long*)y,
                                             gloss, (T*)dlogits, n));
    return check_launch("egk_bce_bwd");
}

int egk_onehot_sigmoid_loss_fwd(egk_stream_t stream, const float* logits, const int64_t* y, float* loss, int32_t rows, int32_t C,
                                int32_t kind, float alpha, float gamma) {
    EGK_REQUIRE(logits && y && loss, "egk_onehot_sigmoid_loss_fwd: null pointer");
    EGK_REQUIRE(C >= 1 && (kind == 0 || kind == 1), "egk_onehot_sigmoid_loss_fwd: bad C / kind");
    const int n = rows * C;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_BCE_FWD, s, 0, 8.0 * n + 8.0 * rows);
    hipLaunchKernelGGL(onehot_sigmoid_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, logits, (const long long*)y, loss, n, C,
                       kind, alpha, gamma);
    return check_launch("egk_onehot_sigmoid_loss_fwd");
}

int egk_onehot_sigmoid_loss_bwd(egk_stream_t stream, const float* logits, const int64_t* y, const float* gloss, void* dlogits,
                                int32_t rows, int32_t C, int32_t kind, float alpha, float gamma, int32_t dtype) {
    EGK_REQUIRE(logits && y && gloss && dlogits, "egk_onehot_sigmoid_loss_bwd: null pointer");
    EGK_REQUIRE(C >= 1 && (kind == 0 || kind == 1), "egk_onehot_sigmoid_loss_bwd: bad C / kind");
    const int n = rows * C;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_BCE_BWD, s, 0, 12.0 * n + 8.0 * rows);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(onehot_sigmoid_bwd_kernel<T>, dim3(cdiv(n, 256)), dim3(256), 0, s, logits,
                                             (const long long*)y, gloss, (T*)dlogits, n, C, kind, alpha, gamma));
    return check_launch("egk_onehot_sigmoid_loss_bwd");
}

int egk_dropout_fwd(egk_stream_t stream, const void* x, void* y, uint8_t* mask, int64_t n, float p, uint64_t seed,
                    uint64_t offset, const uint64_t* dev_offset, int32_t dtype) {
    EGK_REQUIRE(x && y && mask, "egk_dropout_fwd: null pointer");
    EGK_REQUIRE(p >= 0.f && p < 1.f, "egk_dropout_fwd: p out of range");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_DROPOUT_FWD, s, 0, 9.0 * n);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(dropout_fwd_kernel<T>, dim3(ew_grid(n, 4)), dim3(256), 0, s, (const T*)x, (T*)y, mask,
                                             (long long)n, p, seed, offset, dev_offset));
    return check_launch("egk_dropout_fwd");
}

int egk_dropout_bwd(egk_stream_t stream, const void* dy, const uint8_t* mask, void* dx, int64_t n, float p, int32_t dtype) {
    EGK_REQUIRE(dy && mask && dx, "egk_dropout_bwd: null pointer");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_DROPOUT_BWD, s, 0, 9.0 * n);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(dropout_bwd_kernel<T>, dim3(ew_grid(n, 1)), dim3(256), 0, s, (const T*)dy, mask,
                                             (T*)dx, (long long)n, p));
    return check_launch("egk_dropout_bwd");
}

int egk_relu_gate(egk_stream_t stream, const void* dy, const void* y, void* dx, int64_t n, int32_t dtype) {
    EGK_REQUIRE(dy && y && dx, "egk_relu_gate: null pointer");
    if (n == 0) return 0;
    const int vec = (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15) == 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_RELU_GATE, s, 0, 12.0 * n);
    EGK_DISPATCH_T(dtype, hipLaunchKernelGGL(relu_gate_kernel<T>, dim3(ew_grid(n, 4)), dim3(256), 0, s, (const T*)dy, (const T*)y,
                                             (T*)dx, (long long)n, vec));
    return check_launch("egk_relu_gate");
}

int egk_axpby(egk_stream_t stream, const float* x, const float* y, float* out, int64_t n, float a, float b) {
    EGK_REQUIRE(x && out, "egk_axpby: null pointer");
    if (n == 0) return 0;
    const int vec = ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && (!y || ((uintptr_t)y & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_AXPBY, s, 0, (y ? 12.0 : 8.0) * n);
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n, 4)), dim3(256), 0, s, x, y, out, (long long)n, a, b, vec);
    return check_launch("egk_axpby");
}

int egk_fill_scaled(egk_stream_t stream, const float* scalar, float coef, float* out, int64_t n) {
    EGK_REQUIRE(scalar && out, "egk_fill_scaled: null pointer");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_AXPBY, s, 0, 4.0 * n);
    hipLaunchKernelGGL(fill_scaled_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, s, scalar, coef, out, (long long)n);
    return check_launch("egk_fill_scaled");
}

int egk_sum_scale(egk_stream_t stream, const float* x, float* out, int64_t n, float scale, int32_t accumulate) {
    EGK_REQUIRE(x && out, "egk_sum_scale: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_SUM_SCALE, s, 0, 4.0 * n);
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(1024), 0, s, x, out, (long long)n, scale, accumulate);
    return check_launch("egk_sum_scale");
}

int egk_weighted_sums(egk_stream_t stream, const float* const* xs, const int64_t* ns, const float* coefs, int32_t count,
                      float* out) {
    return egk_weighted_sums_acc(stream, xs, ns, coefs, count, out, nullptr);
}

int egk_weighted_sums_acc(egk_stream_t stream, const float* const* xs, const int64_t* ns, const float* coefs, int32_t count,
                          float* out, double* acc) {
    EGK_REQUIRE(xs && ns && coefs && out, "egk_weighted_sums: null pointer");
    EGK_REQUIRE(count >= 1 && count <= MAXVEC, "egk_weighted_sums: 1..%d vectors", MAXVEC);
    VecList v{};
    double bytes = 0;
    for (int k = 0; k < count; ++k) {
        EGK_REQUIRE(xs[k] || ns[k] == 0, "egk_weighted_sums: null vector");
        v.x[k] = xs[k]; v.n[k] = ns[k]; v.coef[k] = coefs[k];
        bytes += 4.0 * ns[k];
    }
    v.count = count;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(KID_SUM_SCALE, s, 0, bytes);
    hipLaunchKernelGGL(weighted_sums_kernel, dim3(1), dim3(1024), 0, s, v, out, acc);
    return check_launch("egk_weighted_sums");
}

int egk_fill_scaled_multi(egk_stream_t stream, const float* scalar, const float* coefs, float* const* outs, const int64_t* ns,
                          int32_t count) {
    EGK_REQUIRE(scalar && coefs && outs && ns, "egk_fill_scaled_multi: null pointer");
    EGK_REQUIRE(count >= 1 && count <= MAXVEC, "egk_fill_scaled_multi: 1..%d vectors", MAXVEC);
    VecList v{};
    long long nmax = 0;
    for (int k = 0; k < count; ++k) {
        EGK_REQUIRE(outs[k] || ns[k] == 0, "egk_fill_scaled_multi: null vector");
        v.out[k] = outs[k]; v.n[k] = ns[k]; v.coef[k] = coefs[k];
        nmax = ns[k] > nmax ? ns[k] : nmax;
    }
    v.count = count;
    if (nmax == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long blocks = (nmax + 255) / 256;
    hipLaunchKernelGGL(fill_scaled_multi_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks), count), dim3(256), 0, s, scalar, v);
    return check_launch("egk_fill_scaled_multi");
}

int egk_copy_blocks(egk_stream_t stream, const void* const* srcs, const int64_t* nbytes, void* dst, int32_t count) {
    EGK_REQUIRE(srcs && nbytes && dst, "egk_copy_blocks: null pointer");
    EGK_REQUIRE(count >= 1 && count <= MAXVEC, "egk_copy_blocks: 1..%d blocks", MAXVEC);
    BlockList b{};
    long long off = 0, nmax = 0;
    for (int k = 0; k < count; ++k) {
        b.src[k] = (const unsigned char*)srcs[k]; b.off[k] = off; b.bytes[k] = nbytes[k];
        off += nbytes[k];
        nmax = nbytes[k] > nmax ? nbytes[k] : nmax;
    }
    b.count = count;
    if (nmax == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long blocks = (nmax / 16 + 255) / 256 + 1;
    hipLaunchKernelGGL(copy_blocks_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks), count), dim3(256), 0, s, b, (unsigned char*)dst);
    return check_launch("egk_copy_blocks");
}

// the flat gradient buffer cleared by a launch of the library (16-byte stores; the buffer is 16-byte aligned and padded)
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* __restrict__ p, long long n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) p[i] = z;
}

int egk_zero_fill(egk_stream_t stream, void* p, int64_t bytes) {
    EGK_REQUIRE(p || bytes == 0, "egk_zero_fill: null pointer");
    EGK_REQUIRE(((uintptr_t)p & 15) == 0 && (bytes & 15) == 0 && bytes >= 0, "egk_zero_fill: 16-byte aligned buffer of whole 16-byte groups");
    if (bytes == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long n16 = bytes / 16;
    const long long blocks = (n16 + 255) / 256;
    // A GENTLE fill: the captured step clears the gradient buffer beside its forward pass, with ~0.4 ms to spare -- at full
    // rate (4096 workgroups, 5 TB/s) the 100 MB burst doubled the HBM-bound row kernel it ran beside (positional-encoding add
    // 7.2 -> 15.7 us, profiles/r05_c3_replay_timeline.txt at 141 us); egk_tune(5, n) sets the workgroup cap.
    // Same box, three alternating rounds of the headline step: 4096 workgroups 1.407-1.417 ms, 192: 1.399-1.413, 64: 1.401-1.408
    const long long cap = g_zero_fill_blocks > 0 ? g_zero_fill_blocks : 64;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(blocks > cap ? cap : blocks)), dim3(256), 0, s, (uint4*)p, n16);
    return check_launch("egk_zero_fill");
}

// up to ZERO_MAX_RANGES byte ranges of one buffer cleared by ONE launch (blockIdx.y = range): the gradient slots that are still
// accumulated into once the matrices whose ONE weight-gradient launch stores its result are left alone (FlatAdam.store_slots)
constexpr int ZERO_MAX_RANGES = 48;
struct ByteRanges {
    long long begin[ZERO_MAX_RANGES], len[ZERO_MAX_RANGES];
};
__global__ __launch_bounds__(256) void zero_ranges_kernel(unsigned char* __restrict__ base, const ByteRanges R) {
    uint4* p = reinterpret_cast<uint4*>(base + R.begin[blockIdx.y]);
    const long long n16 = R.len[blockIdx.y] >> 4;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) p[i] = z;
}

int egk_zero_fill_ranges(egk_stream_t stream, void* base, const int64_t* begin, const int64_t* bytes, int32_t n_ranges) {
    EGK_REQUIRE(base && begin && bytes, "egk_zero_fill_ranges: null pointer");
    EGK_REQUIRE(n_ranges >= 1 && n_ranges <= ZERO_MAX_RANGES, "egk_zero_fill_ranges: 1 .. %d ranges per launch", ZERO_MAX_RANGES);
    EGK_REQUIRE(((uintptr_t)base & 15) == 0, "egk_zero_fill_ranges: 16-byte aligned buffer");
    ByteRanges R;
    long long longest = 0;
    for (int i = 0; i < ZERO_MAX_RANGES; ++i) {
        R.begin[i] = i < n_ranges ? begin[i] : 0;
        R.len[i] = i < n_ranges ? bytes[i] : 0;
        if (i < n_ranges) {
            EGK_REQUIRE(begin[i] >= 0 && bytes[i] >= 0 && begin[i] % 16 == 0 && bytes[i] % 16 == 0,
                        "egk_zero_fill_ranges: ranges of whole 16-byte groups");
            longest = bytes[i] > longest ? bytes[i] : longest;
        }
    }
    if (longest == 0) return 0;
    long long gx = (longest / 16 + 255) / 256;
    if (gx > 16) gx = 16;  // (gentle, as egk_zero_fill: it runs beside the forward pass)
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)gx, n_ranges), dim3(256), 0, (hipStream_t)stream, (unsigned char*)base, R);
    return check_launch("egk_zero_fill_ranges");
}

int egk_adam_hyper(egk_stream_t stream, const float* src, int64_t* t_dev, double beta1, double beta2, float* hyper) {
    EGK_REQUIRE(src && t_dev && hyper, "egk_adam_hyper: null pointer");
    hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, (long long*)t_dev, beta1, beta2, hyper);
    return check_launch("egk_adam_hyper");
}

int egk_adam_step(egk_stream_t stream, float* p, const void* g, int32_t g_dtype, float* m, float* v, int64_t n,
                  const float* hyper, float beta1, float beta2, float eps, float weight_decay, void* bf16_shadow) {
    return egk_adam_step_bump(stream, p, g, g_dtype, m, v, n, hyper, beta1, beta2, eps, weight_decay, bf16_shadow, nullptr, nullptr, 0);
}

int egk_adam_step_bump(egk_stream_t stream, float* p, const void* g, int32_t g_dtype, float* m, float* v, int64_t n,
                       const float* hyper, float beta1, float beta2, float eps, float weight_decay, void* bf16_shadow,
                       void* bf16_lo_shadow, int64_t* bump_word, int64_t bump) {
    EGK_REQUIRE(p && g && m && v && hyper, "egk_adam_step: null pointer");
    EGK_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "egk_adam_step: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    EGK_REQUIRE(!bf16_shadow || ((uintptr_t)bf16_shadow & 7) == 0, "egk_adam_step: shadow must be 8-byte aligned");
    EGK_REQUIRE(!bf16_lo_shadow || ((uintptr_t)bf16_lo_shadow & 7) == 0, "egk_adam_step: low-half shadow must be 8-byte aligned");
    ProfScope prof(KID_ADAM, s, 0, ((bf16_shadow ? 26.0 : 24.0) + (bf16_lo_shadow ? 2.0 : 0.0) + (g_dtype == EGK_BF16 ? 2.0 : 4.0)) * n);
    // Workgroups: one 1024-element group per workgroup up to 32768 of them (egk_tune(6, n) sets the cap).  An optimizer slice runs
    // BESIDE weight-gradient launches in every captured step's tail; round 5 first capped it at 4096 (a 8 us reduction queued beside
    // it had waited 135 us for wave slots) and, once the gradient buffer was no longer cleared and read back, measured the wide grid
    // ahead again: headline 1.365-1.379 against 1.374-1.386 ms, config 4 2.180-2.182 against 2.191-2.206, Hp = 4096 2.722 against
    // 2.751 (narrower is clearly worse: 1024 workgroups 1.405-1.414, 512 1.447)
    const long long want = (n / 4 + 255) / 256;
    const long long cap = g_adam_blocks > 0 ? g_adam_blocks : 32768;
    const unsigned grid = (unsigned)(want < 1 ? 1 : want > cap ? cap : want);
    EGK_DISPATCH_T(g_dtype, hipLaunchKernelGGL(adam_kernel<T>, dim3(grid), dim3(256), 0, s, p, (const T*)g, m, v,
                                               (long long)n, hyper, beta1, beta2, eps, weight_decay, (bf16_t*)bf16_shadow,
                                               (bf16_t*)bf16_lo_shadow, (long long*)bump_word, (long long)bump));
    return check_launch("egk_adam_step");
}
}
