// Experiment (not part of the library): what limits a one-wave-per-row streaming kernel at [6144, 1024] bf16?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/rowcopy tools/exp/rowcopy.hip && /tmp/rowcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short bf16_t;
__device__ inline float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ inline bf16_t f2bf(float f) { unsigned u = __float_as_uint(f); return (bf16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

// (1) wave per row, 4 x 8-byte loads per lane, column = (i*64 + lane)*4
__global__ __launch_bounds__(256) void k_row8(const bf16_t* x, bf16_t* y, int rows, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        uint2 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const uint2*>(x + (long long)row * cols + (i * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = __uint_as_float(v[i].x << 16) * 2.f, b = __uint_as_float(v[i].x & 0xffff0000u) * 2.f;
            float c = __uint_as_float(v[i].y << 16) * 2.f, d = __uint_as_float(v[i].y & 0xffff0000u) * 2.f;
            uint2 o; o.x = f2bf(a) | ((unsigned)f2bf(b) << 16); o.y = f2bf(c) | ((unsigned)f2bf(d) << 16);
            *reinterpret_cast<uint2*>(y + (long long)row * cols + (i * 64 + lane) * 4) = o;
        }
    }
}
// (2) wave per row, 2 x 16-byte loads per lane, column = (i*64 + lane)*8
__global__ __launch_bounds__(256) void k_row16(const bf16_t* x, bf16_t* y, int rows, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        uint4 v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = *reinterpret_cast<const uint4*>(x + (long long)row * cols + (i * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            unsigned o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float a = __uint_as_float(w[t] << 16) * 2.f, b = __uint_as_float(w[t] & 0xffff0000u) * 2.f;
                o[t] = f2bf(a) | ((unsigned)f2bf(b) << 16);
            }
            *reinterpret_cast<uint4*>(y + (long long)row * cols + (i * 64 + lane) * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}
// (3) flat: one 16-byte element per thread, no loop
__global__ __launch_bounds__(256) void k_flat16(const bf16_t* x, bf16_t* y, long long n8) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    uint4 v = reinterpret_cast<const uint4*>(x)[i];
    unsigned w[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float a = __uint_as_float(w[t] << 16) * 2.f, b = __uint_as_float(w[t] & 0xffff0000u) * 2.f;
        o[t] = f2bf(a) | ((unsigned)f2bf(b) << 16);
    }
    reinterpret_cast<uint4*>(y)[i] = make_uint4(o[0], o[1], o[2], o[3]);
}
// (4) wave per row with a wave reduction in the middle (the LayerNorm dependency shape)
__global__ __launch_bounds__(256) void k_row8_red(const bf16_t* x, bf16_t* y, int rows, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        uint2 v[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = *reinterpret_cast<const uint2*>(x + (long long)row * cols + (i * 64 + lane) * 4);
            s += __uint_as_float(v[i].x << 16) + __uint_as_float(v[i].y << 16);
        }
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = __uint_as_float(v[i].x << 16) - s, b = __uint_as_float(v[i].x & 0xffff0000u) - s;
            float c = __uint_as_float(v[i].y << 16) - s, d = __uint_as_float(v[i].y & 0xffff0000u) - s;
            uint2 o; o.x = f2bf(a) | ((unsigned)f2bf(b) << 16); o.y = f2bf(c) | ((unsigned)f2bf(d) << 16);
            *reinterpret_cast<uint2*>(y + (long long)row * cols + (i * 64 + lane) * 4) = o;
        }
    }
}
// (5) the library's shape: wave per row, 8-byte loads, f32 affine rows (w, b) in registers, mean / variance reductions
__global__ __launch_bounds__(256) void k_rowln(const bf16_t* x, const float* w, const float* b, bf16_t* y, int rows, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 wv[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wv[i] = *reinterpret_cast<const float4*>(w + (i * 64 + lane) * 4);
        bv[i] = *reinterpret_cast<const float4*>(b + (i * 64 + lane) * 4);
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        uint2 v[4];
        float f[16], s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const uint2*>(x + (long long)row * cols + (i * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[4 * i] = __uint_as_float(v[i].x << 16); f[4 * i + 1] = __uint_as_float(v[i].x & 0xffff0000u);
            f[4 * i + 2] = __uint_as_float(v[i].y << 16); f[4 * i + 3] = __uint_as_float(v[i].y & 0xffff0000u);
            s += (f[4 * i] + f[4 * i + 1]) + (f[4 * i + 2] + f[4 * i + 3]);
        }
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        const float mu = s / cols;
#pragma unroll
        for (int k = 0; k < 16; ++k) q += (f[k] - mu) * (f[k] - mu);
        for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o);
        const float rs = rsqrtf(q / cols + 1e-5f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = fmaxf((f[4 * i] - mu) * rs * wv[i].x + bv[i].x, 0.f), bb = fmaxf((f[4 * i + 1] - mu) * rs * wv[i].y + bv[i].y, 0.f);
            const float c = fmaxf((f[4 * i + 2] - mu) * rs * wv[i].z + bv[i].z, 0.f), d = fmaxf((f[4 * i + 3] - mu) * rs * wv[i].w + bv[i].w, 0.f);
            uint2 o; o.x = f2bf(a) | ((unsigned)f2bf(bb) << 16); o.y = f2bf(c) | ((unsigned)f2bf(d) << 16);
            *reinterpret_cast<uint2*>(y + (long long)row * cols + (i * 64 + lane) * 4) = o;
        }
    }
}
// device time per launch of 20 DEPENDENT launches replayed from a hipGraph (how the library's kernels are timed)
template <typename F> float timegraph(F f) {
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    f(st); hipStreamSynchronize(st);
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 20; ++i) f(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / 200;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 200; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / 200;
}
int main() {
    const int rows = 6144, cols = 1024; const long long n = (long long)rows * cols;
    bf16_t *x, *y; hipMalloc(&x, n * 2); hipMalloc(&y, n * 2); hipMemset(x, 0x3f, n * 2);
    for (int grid : {256, 512, 1024, 1536}) {
        printf("grid %4d  row8 %.2f us  row16 %.2f us  row8+reduce %.2f us\n", grid,
               timeit([&] { hipLaunchKernelGGL(k_row8, dim3(grid), dim3(256), 0, 0, x, y, rows, cols); }),
               timeit([&] { hipLaunchKernelGGL(k_row16, dim3(grid), dim3(256), 0, 0, x, y, rows, cols); }),
               timeit([&] { hipLaunchKernelGGL(k_row8_red, dim3(grid), dim3(256), 0, 0, x, y, rows, cols); }));
    }
    float *w, *b; hipMalloc(&w, cols * 4); hipMalloc(&b, cols * 4); hipMemset(w, 0, cols * 4); hipMemset(b, 0, cols * 4);
    printf("in a hipGraph, 20 dependent launches:\n");
    for (int grid : {256, 512, 768, 1024, 1536}) {
        printf("grid %4d  row8 %.2f us  row16 %.2f us  row8+reduce %.2f us  rowln %.2f us\n", grid,
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k_row8, dim3(grid), dim3(256), 0, s, x, y, rows, cols); }),
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k_row16, dim3(grid), dim3(256), 0, s, x, y, rows, cols); }),
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k_row8_red, dim3(grid), dim3(256), 0, s, x, y, rows, cols); }),
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k_rowln, dim3(grid), dim3(256), 0, s, x, w, b, y, rows, cols); }));
    }
    printf("flat16 in graph %.2f us\n", timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k_flat16, dim3((n / 8 + 255) / 256), dim3(256), 0, s, x, y, n / 8); }));
    printf("flat16 %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_flat16, dim3((n / 8 + 255) / 256), dim3(256), 0, 0, x, y, n / 8); }));
    printf("memcpy d2d %.2f us\n", timeit([&] { hipMemcpyAsync(y, x, n * 2, hipMemcpyDeviceToDevice, 0); }));
    return 0;
}
