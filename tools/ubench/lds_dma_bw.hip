// Microbenchmark: L2 -> CU staging bandwidth, LDS-DMA (buffer_load_dwordx4 ... lds) vs VGPR loads (buffer_load_dwordx4),
// on an L2-resident footprint.  hipcc --offload-arch=gfx950 -O3 lds_dma_bw.hip -o lds_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 = LDS-DMA, 1 = VGPR loads
__global__ __launch_bounds__(256) void bw_kernel(const char* src, size_t bytes, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) uint4 lds[4][4 * 64 * 4];       // 4 stages x 16 KiB
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)bytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    // each workgroup walks its own 16 KiB tiles through the footprint
    unsigned off = (unsigned)(((size_t)blockIdx.x * 16384) % bytes) + (unsigned)(wave * 4096 + lane * 16);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)&lds[it & 3][(uwave * 4 + i) * 64], 16, (int)(off + i * 1024), 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // two tiles stay in flight
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + i * 1024), 0, 0);
                acc ^= v;
            }
        }
        off += 16384u * 61u;                                     // stride through the footprint
        if (off >= bytes) off -= (unsigned)bytes * (off / (unsigned)bytes);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE == 0) acc.x ^= lds[0][threadIdx.x].x;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atoi(argv[1]) : 16) << 20;    // footprint MiB (L2 is 4 MiB per XCD, 32 MiB total)
    const int iters = 2000;
    char* d; unsigned* sink;
    hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024, 2048}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(bw_kernel<0>, dim3(wgs), dim3(256), 0, 0, d, bytes, iters, sink);
                else hipLaunchKernelGGL(bw_kernel<1>, dim3(wgs), dim3(256), 0, 0, d, bytes, iters, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("footprint %zu MiB  %4d WGs  %s : %.2f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", bytes >> 20, wgs, mode ? "VGPR loads" : "LDS-DMA   ",
                                (double)wgs * iters * 16384 / (ms * 1e-3) / 1e12, (double)wgs * iters * 16384 / (ms * 1e-3) / 256 / 2.4e9);
            }
        }
    }
    return 0;
}
