// Litmus: does a buffer STORE retire (decrement vmcnt) before an OLDER buffer LOAD that misses to HBM?
// Each lane issues a cold load, then a store to a hot line, then s_waitcnt vmcnt(1) and snapshots the load's destination register.
// If the snapshot still holds the sentinel, the store retired first (vmcnt reached 1 with the load outstanding): loads and stores
// are then NOT ordered on the VM counter and a counted wait may not treat younger stores as "allowed outstanding".
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/vmcnt_order.hip -o /tmp/vmcnt_order && /tmp/vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void litmus(const unsigned* cold, unsigned* hot, size_t cold_words, int iters, unsigned long long* early, unsigned long long* total) {
    const __amdgpu_buffer_rsrc_t csrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(cold), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hsrd = __builtin_amdgcn_make_buffer_rsrc(hot, 0, 0x7ffffff0, 0x00020000);
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    unsigned state = gid * 2654435761u + 12345u;
    const unsigned hot_off = (gid & 1023u) * 4u;
    for (int i = 0; i < iters; ++i) {
        state = state * 1664525u + 1013904223u;
        const unsigned coff = (unsigned)(((size_t)state * 64u) % (cold_words * 4u)) & ~3u;       // a random (cold) line each time
        unsigned ld, snap;
        asm volatile(
            "v_mov_b32 %0, 0xdeadbeef\n\t"
            "s_nop 4\n\t"
            "buffer_load_dword %0, %2, %3, 0 offen\n\t"
            "buffer_store_dword %4, %5, %6, 0 offen\n\t"
            "s_waitcnt vmcnt(1)\n\t"
            "v_mov_b32 %1, %0\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            : "=&v"(ld), "=&v"(snap)
            : "v"(coff), "s"(csrd), "v"(state), "v"(hot_off), "s"(hsrd)
            : "memory");
        bad += (snap == 0xdeadbeefu) ? 1ull : 0ull;
        state ^= ld;
    }
    atomicAdd(early, bad);
    atomicAdd(total, (unsigned long long)iters);
}

// The same question for the LDS-DMA form of the load (what the conv K ring uses): cold `buffer_load_dword ... lds`, hot store,
// s_waitcnt vmcnt(1), then read the LDS word the DMA fills.
__global__ void litmus_lds(const unsigned* cold, unsigned* hot, size_t cold_words, int iters, unsigned long long* early, unsigned long long* total) {
    __shared__ unsigned slot[256];
    const __amdgpu_buffer_rsrc_t csrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(cold), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hsrd = __builtin_amdgcn_make_buffer_rsrc(hot, 0, 0x7ffffff0, 0x00020000);
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned wave = threadIdx.x >> 6;
    const unsigned lds_wave_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)&slot[wave * 64]);   // M0: this wave's 64 words
    const unsigned lds_lane_addr = lds_wave_base + (threadIdx.x & 63) * 4;
    unsigned long long bad = 0;
    unsigned state = gid * 2654435761u + 777u;
    const unsigned hot_off = (gid & 1023u) * 4u;
    for (int i = 0; i < iters; ++i) {
        state = state * 1664525u + 1013904223u;
        const unsigned coff = (unsigned)(((size_t)state * 64u) % (cold_words * 4u)) & ~3u;
        unsigned snap, sentinel = 0xdeadbeefu;
        asm volatile(
            "ds_write_b32 %1, %2\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 1\n\t"
            "buffer_load_dword %4, %5, 0 offen lds\n\t"
            "buffer_store_dword %6, %7, %8, 0 offen\n\t"
            "s_waitcnt vmcnt(1)\n\t"
            "ds_read_b32 %0, %1\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            : "=&v"(snap)
            : "v"(lds_lane_addr), "v"(sentinel), "s"(lds_wave_base), "v"(coff), "s"(csrd), "v"(state), "v"(hot_off), "s"(hsrd)
            : "memory");
        bad += (snap == 0xdeadbeefu) ? 1ull : 0ull;
        state ^= snap;
    }
    atomicAdd(early, bad);
    atomicAdd(total, (unsigned long long)iters);
}

int main() {
    const size_t cold_bytes = (size_t)2 << 30;                         // 2 GiB: far beyond L2 + MALL
    unsigned *cold, *hot;
    unsigned long long *cnt;
    hipMalloc(&cold, cold_bytes); hipMalloc(&hot, 4096); hipMalloc(&cnt, 16);
    hipMemset(cold, 0x11, cold_bytes); hipMemset(hot, 0, 4096); hipMemset(cnt, 0, 16);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(litmus, dim3(256 * 8), dim3(256), 0, 0, cold, hot, cold_bytes / 4, 2000, cnt, cnt + 1);
        hipDeviceSynchronize();
    }
    unsigned long long h[2];
    hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
    printf("VGPR load : store-before-older-load retirements observed: %llu of %llu trials\n", h[0], h[1]);
    hipMemset(cnt, 0, 16);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(litmus_lds, dim3(256 * 8), dim3(256), 0, 0, cold, hot, cold_bytes / 4, 2000, cnt, cnt + 1);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
    printf("LDS-DMA   : store-before-older-load retirements observed: %llu of %llu trials\n", h[0], h[1]);
    return 0;
}
