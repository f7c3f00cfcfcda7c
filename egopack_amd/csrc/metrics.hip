// Validation-side kernels: integer work on logits / predictions that are already resident in
// HBM, so that the validation loop needs no device->host copy of [N, C] logits per batch.
//   egk_label_rank    rank of the ground-truth class inside each logits row (top-k accuracy / recall are counts of
//                     rank < k: reference utils/meters/utils.py:6-28 topk_accuracy, torchmetrics MulticlassAccuracy)
//   egk_edit_distance Levenshtein distance of K sampled label sequences against the ground truth
//                     (reference utils/meters/ego4d.py:410-423 ``editdistance.eval(pred, label) / Z``, min over K on the host)
#include "common.h"

namespace egk {

// one wave per row; rank = #{j : s_j > s_y} + #{j < y : s_j == s_y}  (ties go to the lower class index), -1 when the
// label is negative (ignore_index) or out of range.  NaN scores never outrank anything.
__global__ __launch_bounds__(256) void label_rank_kernel(const float* __restrict__ logits, long long ld,
                                                         const long long* __restrict__ labels, long long label_stride,
                                                         int* __restrict__ rank, int rows, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const long long y = labels[(long long)row * label_stride];
        if (y < 0 || y >= C) {
            if (lane == 0) rank[row] = -1;
            continue;
        }
        const float* r = logits + (long long)row * ld;
        const float sy = r[y];
        int cnt = 0;
        for (int j = lane; j < C; j += 64) {
            const float v = r[j];
            cnt += (v > sy || (v == sy && j < (int)y)) ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) rank[row] = cnt;
    }
}

// one thread per (sequence n, sample k): single-row dynamic programme over Z <= 64 positions.
// pred [N, Z, K] (sample index fastest, as reference predictions.reshape(-1, 22, 5)[:, 2:] slices), label [N, Z]
__global__ __launch_bounds__(64) void edit_distance_kernel(const long long* __restrict__ pred, long long p_sn, long long p_sz,
                                                           long long p_sk, const long long* __restrict__ label,
                                                           long long l_sn, long long l_sz, int* __restrict__ out, int N,
                                                           int Z, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * K) return;
    const int n = idx / K, k = idx % K;
    int prev[65];
    for (int j = 0; j <= Z; ++j) prev[j] = j;
    for (int i = 1; i <= Z; ++i) {
        const long long a = pred[n * p_sn + (long long)(i - 1) * p_sz + k * p_sk];
        int diag = prev[0];
        prev[0] = i;
        for (int j = 1; j <= Z; ++j) {
            const long long b = label[n * l_sn + (long long)(j - 1) * l_sz];
            const int sub = diag + (a == b ? 0 : 1);
            const int del = prev[j] + 1, ins = prev[j - 1] + 1;
            diag = prev[j];
            prev[j] = min(sub, min(del, ins));
        }
    }
    out[idx] = prev[Z];
}

}  // namespace egk

using namespace egk;

extern "C" {

int egk_label_rank(egk_stream_t stream, const float* logits, int64_t ld, const int64_t* labels, int64_t label_stride,
                   int32_t* rank, int32_t rows, int32_t C) {
    EGK_REQUIRE(logits && labels && rank, "egk_label_rank: null pointer");
    EGK_REQUIRE(C >= 1 && ld >= C, "egk_label_rank: bad class count / leading dimension");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int grid = cdiv(rows, 4);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(label_rank_kernel, dim3(grid), dim3(256), 0, s, logits, (long long)ld, (const long long*)labels,
                       (long long)label_stride, rank, rows, C);
    return check_launch("egk_label_rank");
}

int egk_edit_distance(egk_stream_t stream, const int64_t* pred, int64_t p_sn, int64_t p_sz, int64_t p_sk, const int64_t* label,
                      int64_t l_sn, int64_t l_sz, int32_t* out, int32_t N, int32_t Z, int32_t K) {
    EGK_REQUIRE(pred && label && out, "egk_edit_distance: null pointer");
    EGK_REQUIRE(Z >= 0 && Z <= 64, "egk_edit_distance: sequence length %d > 64 unsupported", Z);
    if (N == 0 || K == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(edit_distance_kernel, dim3(cdiv(N * K, 64)), dim3(64), 0, s, (const long long*)pred, (long long)p_sn,
                       (long long)p_sz, (long long)p_sk, (const long long*)label, (long long)l_sn, (long long)l_sz, out, N, Z, K);
    return check_launch("egk_edit_distance");
}

}  // extern "C"
