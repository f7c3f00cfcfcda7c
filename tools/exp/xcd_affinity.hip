// Does a consumer launch read a freshly written tensor faster when its workgroups run on the XCD whose workgroups WROTE the same rows
// in the previous launch?  (MI355X: 8 XCDs, each with its own 4 MiB L2; workgroups are dealt round-robin, b % 8 = XCD.)
// Producer: writes a [6144][1024] bf16 tensor (12.6 MB), 4 rows per 256-thread workgroup.  Consumer: reads it and writes another one
// (the traffic of a row LayerNorm).  Mappings: 0 = block b owns rows 4b.. (rows interleaved over the XCDs every 4 rows: what the row
// kernels do); 1 = XCD x owns the contiguous rows [x * rows / 8, (x + 1) * rows / 8).
// hipcc --offload-arch=gfx950 -O3 tools/exp/xcd_affinity.hip -o tools/exp/build/xcd_affinity && tools/exp/build/xcd_affinity
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ROWS = 6144, COLS = 1024, RPB = 4;

__device__ __forceinline__ int vblock(int b, int nblk, int mode) { return mode ? (b & 7) * (nblk >> 3) + (b >> 3) : b; }

__global__ __launch_bounds__(256) void producer(unsigned short* __restrict__ y, int mode, unsigned seedv) {
    const int v = vblock(blockIdx.x, gridDim.x, mode);
    const int row = v * RPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint4 val = make_uint4(seedv + row, lane, seedv ^ lane, row);
    uint4* p = reinterpret_cast<uint4*>(y + (long long)row * COLS);
    p[lane] = val;
    p[64 + lane] = val;
}

__global__ __launch_bounds__(256) void consumer(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int mode) {
    const int v = vblock(blockIdx.x, gridDim.x, mode);
    const int row = v * RPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint4* p = reinterpret_cast<const uint4*>(x + (long long)row * COLS);
    uint4 a = p[lane], b = p[64 + lane];
    a.x += b.y; a.y ^= b.x; b.z += a.w;
    uint4* q = reinterpret_cast<uint4*>(y + (long long)row * COLS);
    q[lane] = a;
    q[64 + lane] = b;
}

static float loop_us(int pm, int cm, bool with_consumer, unsigned short* a, unsigned short* b, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int nblk = ROWS / RPB;
    for (int w = 0; w < 5; ++w) {
        producer<<<nblk, 256>>>(a, pm, w);
        if (with_consumer) consumer<<<nblk, 256>>>(a, b, cm);
    }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) {
        producer<<<nblk, 256>>>(a, pm, i);
        if (with_consumer) consumer<<<nblk, 256>>>(a, b, cm);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

// the same pairs replayed from a hipGraph (the training step is a captured graph: do its kernel nodes keep the L2 contents too?)
static float graph_us(int pm, int cm, bool with_consumer, unsigned short* a, unsigned short* b, int iters) {
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraph_t gr;
    hipGraphExec_t ex;
    const int nblk = ROWS / RPB;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < iters; ++i) {
        producer<<<nblk, 256, 0, st>>>(a, pm, i);
        if (with_consumer) consumer<<<nblk, 256, 0, st>>>(a, b, cm);
    }
    hipStreamEndCapture(st, &gr);
    hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    hipGraphLaunch(ex, st);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    hipGraphLaunch(ex, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ex); hipGraphDestroy(gr); hipStreamDestroy(st);
    return ms * 1e3f / iters;
}

int main() {
    unsigned short *a, *b;
    hipMalloc(&a, (size_t)ROWS * COLS * 2);
    hipMalloc(&b, (size_t)ROWS * COLS * 2);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep)
        for (int pm = 0; pm < 2; ++pm) {
            const float p = loop_us(pm, 0, false, a, b, iters);
            for (int cm = 0; cm < 2; ++cm) {
                const float pc = loop_us(pm, cm, true, a, b, iters);
                printf("producer map %d (%.2f us alone) -> consumer map %d: pair %.2f us, consumer ~ %.2f us\n", pm, p, cm, pc, pc - p);
            }
        }
    for (int pm = 0; pm < 2; ++pm) {
        const float p = graph_us(pm, 0, false, a, b, iters);
        for (int cm = 0; cm < 2; ++cm) {
            const float pc = graph_us(pm, cm, true, a, b, iters);
            printf("hipGraph: producer map %d (%.2f us alone) -> consumer map %d: pair %.2f us, consumer ~ %.2f us\n", pm, p, cm, pc, pc - p);
        }
    }
    return 0;
}
