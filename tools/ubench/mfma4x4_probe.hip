// Probe: operand / result lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 outer products, K = 1).
//   hipcc --offload-arch=gfx950 -O2 -o mfma4x4_probe mfma4x4_probe.hip && ./mfma4x4_probe
// Feeds A = 100 * lane, B = lane-dependent one-hot patterns and prints which (lane, vgpr) holds which product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, const float* a, const float* b) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
    float ha[64], hb[64], ho[256];
    float *da, *db, *dout;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1024);
    // A[lane] = 1 + lane, B[lane] = 1000 * (1 + lane): product D = A[la] * B[lb] identifies (la, lb) uniquely
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 1000.f * (1 + l); }
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dout, da, db);
    hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const float p = ho[l * 4 + v];
            const int lb = (int)(p / 1000.f + 0.5f) / 1;   // p = (1+la) * 1000 * (1+lb)
            // find (la, lb)
            int fa = -1, fb = -1;
            for (int x = 0; x < 64 && fa < 0; ++x) for (int y = 0; y < 64; ++y) if ((1.f + x) * 1000.f * (1 + y) == p) {
                // candidates are ambiguous (products collide); prefer the hypothesis la = 4*(l/4) + v, lb = l
                if (x == 4 * (l / 4) + v && y == l) { fa = x; fb = y; break; }
            }
            if (fa < 0) { ok = 0; printf("lane %d vgpr %d: value %g does not match hypothesis A-lane %d x B-lane %d (= %g)\n", l, v, p, 4 * (l / 4) + v, l, (1.f + 4 * (l / 4) + v) * 1000.f * (1 + l)); }
            (void)lb;
        }
    printf(ok ? "HYPOTHESIS OK: D[vgpr i] at lane l = A[lane 4*(l/4)+i] * B[lane l]  (block = l/4, row i, col l%%4)\n" : "hypothesis FAILED\n");
    return 0;
}
