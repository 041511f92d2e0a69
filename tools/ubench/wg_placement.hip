// Which workgroups of a 512-workgroup grid (two per CU by LDS) share a CU?  Prints, per workgroup id, the XCC / SE / CU it ran on and
// whether the pairing (b, b + 256) holds.  hipcc --offload-arch=gfx950 -O3 wg_placement.hip -o wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
    __shared__ uint4 lds[73856 / 16];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = make_uint4(hwid, xcc, 0, 0);
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);    // keep every workgroup resident while the grid fills
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = lds[0].x; out[blockIdx.x * 2 + 1] = lds[0].y; }
}
int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 512;
    unsigned* d; hipMalloc(&d, grid * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 2000);
        std::vector<unsigned> h(grid * 2);
        hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> cu;
        for (int b = 0; b < grid; ++b) {
            const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
            const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            cu[(xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(b);
        }
        int pair_half = 0, pair_adj = 0, n2 = 0;
        for (auto& kv : cu) if (kv.second.size() == 2) { ++n2; const int a = kv.second[0], b = kv.second[1]; if (b - a == grid / 2) ++pair_half; if ((b - a) == 8) ++pair_adj; }
        printf("rep %d: %zu distinct CUs, %d with two workgroups; pairs (b, b + grid/2): %d, pairs (b, b + 8): %d\n", rep, cu.size(), n2, pair_half, pair_adj);
        if (rep == 0) { int shown = 0; for (auto& kv : cu) { if (shown++ < 12) { printf("  cu %06x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); } } }
    }
    return 0;
}
