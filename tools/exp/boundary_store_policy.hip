// What does a dependent kernel boundary cost behind a kernel that leaves 12.6 MB of freshly written output, by STORE POLICY of
// that kernel?  (MI355X guide, price-list row "boundary": + B / 6 TB/s when the predecessor leaves B bytes dirty in the XCD
// L2s; "publish-large": write-through sc1 stores leave nothing to write back.)  Chain: W (writes 6144 x 1024 bf16 rows, 16 B per
// lane) -> R (reads them all, 16 B per lane, folds them into 1 KB) x PAIRS, captured in one hipGraph; time per pair for plain,
// nt and sc1 stores.  hipcc -O3 --offload-arch=gfx950 tools/exp/boundary_store_policy.hip -o tools/exp/build/boundary_store_policy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void writer(uint4* __restrict__ dst, const uint4* __restrict__ src, long long n16, unsigned salt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        uint4 v = src[i];
        v.x += salt; v.y ^= salt;
        uint4* p = dst + i;
        if (MODE == 0) *p = v;
        else if (MODE == 1) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
        }
        else {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w = {v.x, v.y, v.z, v.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
        }
    }
}
__global__ __launch_bounds__(256) void reader(const uint4* __restrict__ src, uint4* __restrict__ out, long long n16) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        const uint4 v = src[i];
        acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) out[threadIdx.x] = acc;  // (practically never: keeps the loads alive)
}

int main() {
    const long long rows = 6144, cols = 1024, bytes = rows * cols * 2, n16 = bytes / 16;
    uint4 *a, *b, *c, *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(c, 3, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int PAIRS = 40, GRID = 1536;
    const char* names[3] = {"plain", "nt", "sc1"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < PAIRS; ++p) {
            uint4* dst = (p & 1) ? b : c;
            if (mode == 0) hipLaunchKernelGGL(writer<0>, dim3(GRID), dim3(256), 0, s, dst, a, n16, (unsigned)p);
            else if (mode == 1) hipLaunchKernelGGL(writer<1>, dim3(GRID), dim3(256), 0, s, dst, a, n16, (unsigned)p);
            else hipLaunchKernelGGL(writer<2>, dim3(GRID), dim3(256), 0, s, dst, a, n16, (unsigned)p);
            hipLaunchKernelGGL(reader, dim3(GRID), dim3(256), 0, s, dst, out, n16);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("%-6s stores: %.2f us per (write 12.6 MB -> read 12.6 MB) pair\n", names[mode], best * 1e3f / PAIRS);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
