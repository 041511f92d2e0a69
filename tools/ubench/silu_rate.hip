// Microbenchmark (round 6): what a SiLU costs on gfx950's vector pipes, per 64 values, in shader cycles -- the current form
// x * rcp(1 + exp2(-x log2 e)) in f32, the same with f16 transcendentals, and the raw issue rates of v_exp_f32 / v_rcp_f32 / v_exp_f16 /
// v_rcp_f16 / v_fma_f32 / v_pk_fma_f32 / v_pk_fma_f16.  One wave per SIMD (256 threads per workgroup, one workgroup per CU) and four
// waves per SIMD.   hipcc -O3 --offload-arch=gfx950 silu_rate.hip -o silu_rate && ./silu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP 64
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x % 13) * 0.37f - 2.0f + i * 0.01f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = x[i];
                if (MODE == 0) { asm volatile("v_exp_f32 %0, %0" : "+v"(v)); }
                else if (MODE == 1) { asm volatile("v_rcp_f32 %0, %0" : "+v"(v)); }
                else if (MODE == 2) { asm volatile("v_exp_f16 %0, %0" : "+v"(v)); }
                else if (MODE == 3) { asm volatile("v_rcp_f16 %0, %0" : "+v"(v)); }
                else if (MODE == 4) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v)); }
                else if (MODE == 5) { v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); asm volatile("" : "+v"(v)); }
                else if (MODE == 6) {   // f16 transcendentals on one value: cvt, exp, add, rcp, cvt, mul
                    float t = v * -1.4426950408889634f;
                    _Float16 h = (_Float16)t;
                    asm volatile("v_exp_f16 %0, %0" : "+v"(h));
                    h = h + (_Float16)1.0f;
                    asm volatile("v_rcp_f16 %0, %0" : "+v"(h));
                    v = v * (float)h;
                    asm volatile("" : "+v"(v));
                }
                x[i] = v;
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;      // every wave: the oldest wave of a SIMD is served first, the LAST one to finish is the throughput
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void kp(float* out, long long* cyc, int iters) {
    f2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (f2){(float)(threadIdx.x % 13) * 0.37f - 2.0f + i * 0.01f, 0.5f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
                else asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(x[i].x));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;      // every wave: the oldest wave of a SIMD is served first, the LAST one to finish is the throughput
}

template <class K>
static void run(const char* name, K kern, int threads, int vals) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8); hipMemset(cyc, 0, 256 * 16 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 16);
    hipMemcpy(h.data(), cyc, 256 * 16 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < 256; ++b) { long long m = 0; for (int w = 0; w < threads / 64; ++w) m = std::max(m, h[b * 16 + w]); avg += (double)m; }
    avg /= 256;
    const int waves_per_simd = threads / 256;
    // cycles per wave-instruction (or per SiLU of 64 values) as seen by ONE SIMD: elapsed / (instructions per wave x waves on the SIMD)
    printf("%-34s %4d threads: %.2f cycles per wave-op per SIMD (%d value(s) per lane and op)\n", name, threads, avg / ((double)iters * REP * waves_per_simd), vals);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run("v_exp_f32", k<0>, threads, 1); run("v_rcp_f32", k<1>, threads, 1); run("v_exp_f16", k<2>, threads, 1); run("v_rcp_f16", k<3>, threads, 1);
        run("v_fma_f32", k<4>, threads, 1); run("v_pk_fma_f32", kp<0>, threads, 2); run("v_pk_fma_f16", kp<1>, threads, 2);
        run("SiLU f32 (x*rcp(1+exp(-x)))", k<5>, threads, 1); run("SiLU with f16 exp / rcp", k<6>, threads, 1);
    }
    return 0;
}
