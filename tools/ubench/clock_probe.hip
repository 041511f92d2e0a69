// Probe: the shader clock the chip SUSTAINS while every SIMD issues MFMAs back to back -- the denominator of "MFMA utilisation".
//   hipcc --offload-arch=gfx950 -O2 -o clock_probe clock_probe.hip && ./clock_probe
// s_memtime counts shader-clock cycles, s_memrealtime the constant 100 MHz reference: their ratio over a kernel's life is the
// clock that kernel ran at.  Three loads: (a) dense v_mfma_f32_16x16x32_bf16 on every SIMD (2 waves each), (b) the same with a
// v_exp_f32 / v_rcp_f32 pair per MFMA (a SiLU-bound mix), (c) a single wavefront (idle chip).  The kernel also reports the MFMA
// issue rate it reached (cycles per MFMA and wave), so the 16-cycle figure behind the 2.5 PFLOP/s peak is checked in the same run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void load_kernel(unsigned long long* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float e = seed;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) e = e * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e));
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = e;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (threadIdx.x % 64 == 0) {
        unsigned long long* o = out + ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 4;
        o[0] = c1 - c0; o[1] = r1 - r0; o[2] = (unsigned long long)(s != 12345.f);
    }
}

template <int MODE>
static void run(const char* what, int grid, int block, int iters) {
    unsigned long long* d;
    const int waves = grid * block / 64;
    hipMalloc(&d, (size_t)waves * 32);
    load_kernel<MODE><<<grid, block>>>(d, 100, 1.0f);     // warm-up
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    load_kernel<MODE><<<grid, block>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)waves * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cpm;
    for (int w = 0; w < waves; ++w) {
        const double cyc = (double)h[w * 4], ref = (double)h[w * 4 + 1];
        if (ref > 0) { ghz.push_back(cyc / (ref * 10.0)); cpm.push_back(cyc / (8.0 * iters)); }     // ref ticks are 10 ns
    }
    std::sort(ghz.begin(), ghz.end()); std::sort(cpm.begin(), cpm.end());
    const double flops = 2.0 * 16 * 16 * 32 * 8.0 * iters * waves;
    printf("%-34s %5d waves  %.3f ms  shader clock %.3f GHz (min %.3f max %.3f)  cycles per MFMA and wave %.2f  %.0f TFLOP/s\n", what, waves, ms,
           ghz[ghz.size() / 2], ghz.front(), ghz.back(), cpm[cpm.size() / 2], flops / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    run<0>("one wave, MFMA only", 1, 64, iters);
    run<0>("256 CUs x 4 waves, MFMA only", 256, 256, iters);
    run<0>("256 CUs x 8 waves, MFMA only", 256, 512, iters);
    run<1>("256 CUs x 8 waves, MFMA + exp + rcp", 256, 512, iters);
    run<0>("256 CUs x 8 waves, MFMA only (2nd)", 256, 512, iters * 4);
    return 0;
}
