// Probe (round 6): does the order in which a consumer reads a tensor its producer has just written matter?  A kernel writes a buffer of S MB
// front to back (workgroup b writes chunk b, persistent walk), a second kernel reads it front to back or back to front; buffers larger than
// the caches behind L2 (MI355X: 256 MB Infinity Cache) lose their first-written part before the reader gets there, unless it starts at the end.
//   hipcc -O3 --offload-arch=gfx950 mall_order.hip -o mall_order && ./mall_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 64 * 1024;                       // bytes per workgroup step (256 threads x 16 B x 16)

__global__ __launch_bounds__(256) void wr(u32x4* p, long nchunks, unsigned v) {
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        u32x4* q = p + c * (CHUNK / 16);
#pragma unroll
        for (int i = 0; i < 16; ++i) q[i * 256 + threadIdx.x] = (u32x4){v, v + 1, v + 2, (unsigned)c};
    }
}
__global__ __launch_bounds__(256) void rd(const u32x4* p, long nchunks, int rev, unsigned* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long cc = rev ? nchunks - 1 - c : c;
        const u32x4* q = p + cc * (CHUNK / 16);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= q[i * 256 + threadIdx.x];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}
int main() {
    unsigned* sink; hipMalloc(&sink, 4);
    for (long mb : {64, 128, 210, 420, 840, 1680}) {
        const long bytes = mb << 20, nch = bytes / CHUNK;
        u32x4* p; hipMalloc(&p, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rev = 0; rev < 2; ++rev) {
            float best = 1e9f, bw = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                wr<<<1024, 256>>>(p, nch, rep);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float wms; hipEventElapsedTime(&wms, e0, e1);
                hipEventRecord(e0);
                rd<<<1024, 256>>>(p, nch, rev, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; bw = wms; }
            }
            printf("%5ld MB  read %s: %.3f ms = %.2f TB/s   (write before it %.3f ms = %.2f TB/s)\n", mb, rev ? "back to front" : "front to back", best, bytes / best / 1e9, bw, bytes / bw / 1e9);
        }
        hipFree(p);
    }
    return 0;
}
