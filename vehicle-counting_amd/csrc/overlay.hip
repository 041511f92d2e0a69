// Visualisation egress: the annotation overlay of /root/reference/utilities/counting/utils.py:299-331
// (visualize_merged -> draw_anno :104-121, visualize_one_frame :250-274 -> draw_start_last_points :7-15 + draw_one_box :17-34,
// draw_text :36-102, draw_frame_count :123-126) drawn into BGR u8 frames that are already in HBM, off the hot path.
//
// The reference draws with OpenCV (anti-aliased Hershey fonts, cv2.line / circle / rectangle); OpenCV is not in this image, so
// pixel parity with it is UNPINNED.  What is pinned is the primitive list (which shapes, where, which colour, in which order: the
// host side in overlay.py follows the reference call by call) and the rasteriser below against a NumPy restatement of the same
// integer rules (tests/test_gpu_overlay.py).  Text arrives as glyph primitives carrying their 5 x 7 bitmap, so the device holds no font.
//
// One workgroup per frame walks the frame's primitives IN ORDER (painter's algorithm, a workgroup barrier between primitives):
// the result does not depend on scheduling.  Rules (all integer; fdiv = floor division):
//   LINE  (x0,y0)-(x1,y1), thickness t: n = max(|dx|,|dy|); point i = (x0 + fdiv(2*dx*i + n, 2n), y0 + fdiv(2*dy*i + n, 2n)), i = 0..n
//         (n = 0: the single point); every point paints the t x t square whose top-left corner is (x - t/2, y - t/2).
//   DISC  centre (x0,y0), radius t: pixels with dx*dx + dy*dy <= t*t.
//   RECT  outline with corners (x0,y0),(x1,y1), thickness t: its four LINEs.      FILL: the closed box between the corners.
//   GLYPH top-left (x0,y0), scale t, 35 bitmap bits (bit 5*row + col, rows top to bottom): set bits paint a t x t block.
#include "engine.h"

namespace vc {

enum { OV_LINE = 0, OV_DISC = 1, OV_RECT = 2, OV_FILL = 3, OV_GLYPH = 4 };
struct OvPrim { int type, x0, y0, x1, y1, t, color, bits_lo, bits_hi, pad0, pad1, pad2; };     // 12 x int32, matches overlay.py

__device__ __forceinline__ int ov_fdiv(long long a, long long b) { return (int)((a >= 0) ? a / b : -((-a + b - 1) / b)); }   // b > 0

__device__ __forceinline__ void ov_put(uint8_t* f, int H, int W, int x, int y, int color) {
    if (x < 0 || y < 0 || x >= W || y >= H) return;
    uint8_t* p = f + ((size_t)y * W + x) * 3;
    p[0] = (uint8_t)(color & 255); p[1] = (uint8_t)((color >> 8) & 255); p[2] = (uint8_t)((color >> 16) & 255);
}

__device__ void ov_line(uint8_t* f, int H, int W, int x0, int y0, int x1, int y1, int t, int color) {
    const int dx = x1 - x0, dy = y1 - y0, n = max(abs(dx), abs(dy));
    const int t2 = t * t;
    for (int w = threadIdx.x; w < (n + 1) * t2; w += blockDim.x) {
        const int i = w / t2, s = w - i * t2;
        const int x = n ? x0 + ov_fdiv(2ll * dx * i + n, 2ll * n) : x0, y = n ? y0 + ov_fdiv(2ll * dy * i + n, 2ll * n) : y0;
        ov_put(f, H, W, x - t / 2 + s % t, y - t / 2 + s / t, color);
    }
}

__global__ __launch_bounds__(256) void overlay_kernel(uint8_t* frames, int H, int W, const OvPrim* prims, const int* first) {
    uint8_t* f = frames + (size_t)blockIdx.x * H * W * 3;
    for (int k = first[blockIdx.x]; k < first[blockIdx.x + 1]; ++k) {
        const OvPrim p = prims[k];
        if (p.type == OV_LINE) {
            ov_line(f, H, W, p.x0, p.y0, p.x1, p.y1, max(p.t, 1), p.color);
        } else if (p.type == OV_RECT) {
            const int t = max(p.t, 1);
            ov_line(f, H, W, p.x0, p.y0, p.x1, p.y0, t, p.color);
            ov_line(f, H, W, p.x1, p.y0, p.x1, p.y1, t, p.color);
            ov_line(f, H, W, p.x1, p.y1, p.x0, p.y1, t, p.color);
            ov_line(f, H, W, p.x0, p.y1, p.x0, p.y0, t, p.color);
        } else if (p.type == OV_DISC) {
            const int r = max(p.t, 0), d = 2 * r + 1;
            for (int w = threadIdx.x; w < d * d; w += blockDim.x) {
                const int ox = w % d - r, oy = w / d - r;
                if (ox * ox + oy * oy <= r * r) ov_put(f, H, W, p.x0 + ox, p.y0 + oy, p.color);
            }
        } else if (p.type == OV_FILL) {
            const int xa = max(min(p.x0, p.x1), 0), xb = min(max(p.x0, p.x1), W - 1), ya = max(min(p.y0, p.y1), 0), yb = min(max(p.y0, p.y1), H - 1);
            const int bw = xb - xa + 1, bh = yb - ya + 1;
            if (bw > 0 && bh > 0)
                for (int w = threadIdx.x; w < bw * bh; w += blockDim.x) ov_put(f, H, W, xa + w % bw, ya + w / bw, p.color);
        } else if (p.type == OV_GLYPH) {
            const int s = max(p.t, 1), cell = s * s;
            for (int w = threadIdx.x; w < 35 * cell; w += blockDim.x) {
                const int bit = w / cell, q = w - bit * cell;
                const bool on = bit < 32 ? (p.bits_lo >> bit) & 1 : (p.bits_hi >> (bit - 32)) & 1;
                if (on) ov_put(f, H, W, p.x0 + (bit % 5) * s + q % s, p.y0 + (bit / 5) * s + q / s, p.color);
            }
        }
        __syncthreads();                                     // painter's order: the next primitive paints over this one
    }
}

}  // namespace vc

using namespace vc;

extern "C" int vc_overlay(vc_engine* e, void* frames_dev, int b, int h, int w, const int32_t* prims12, const int32_t* frame_first) {
    VC_CHECK(e && frames_dev && prims12 && frame_first && b >= 1 && h >= 1 && w >= 1, VC_ERR_ARG, "bad argument");
    VC_CHECK(frame_first[0] == 0, VC_ERR_ARG, "frame_first[0] must be 0");
    for (int i = 0; i < b; ++i) VC_CHECK(frame_first[i + 1] >= frame_first[i], VC_ERR_ARG, "frame_first must not decrease");
    const int n = frame_first[b];
    if (n == 0) return VC_OK;
    VC_HIP(hipSetDevice(e->cfg.device));
    const size_t need = (size_t)n * sizeof(OvPrim) + (size_t)(b + 1) * sizeof(int);
    if (need > e->overlay_bytes) {
        VC_TRY(dev_alloc(e, (void**)&e->d_overlay, need * 2));
        e->overlay_bytes = need * 2;
    }
    OvPrim* d_prims = (OvPrim*)e->d_overlay;
    int* d_first = (int*)((char*)e->d_overlay + (size_t)n * sizeof(OvPrim));
    VC_HIP(hipMemcpyAsync(d_prims, prims12, (size_t)n * sizeof(OvPrim), hipMemcpyHostToDevice, e->stream));
    VC_HIP(hipMemcpyAsync(d_first, frame_first, (size_t)(b + 1) * sizeof(int), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(overlay_kernel, dim3(b), dim3(256), 0, e->stream, (uint8_t*)frames_dev, h, w, d_prims, d_first);
    VC_HIP(hipGetLastError());
    VC_HIP(hipStreamSynchronize(e->stream));
    return VC_OK;
}
