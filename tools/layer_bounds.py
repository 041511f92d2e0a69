"""Per-launch bounds of a conv_breakdown table: for every conv launch of a step its matrix, HBM and SiLU floors next to the measured time.

    python tools/layer_bounds.py profiles/r05_conv_layers_s640.txt > profiles/r05_layer_bounds.md

MFMA floor = FLOPs / 2.5 PFLOP/s (dense bf16 peak); HBM floor = (input + weights + output bytes of the launch as it runs: fused kernels count only
what leaves the CU) / 6.3 TB/s (the achievable rate of MI355X_MICROARCH.md); SiLU floor = activation evaluations x 27 SIMD cycles per 64 values
(measured: tools/ubench/silu_rate.hip, x * rcp(1 + exp2(-x log2 e)) at four waves per SIMD -- v_exp_f32 / v_rcp_f32 7.7 cycles each, a plain f32
operation 2.6; rounds 1 - 5 priced it at 52 from two quarter-rate transcendentals) / 1024 SIMDs / 2.3 GHz.  ReLU / linear epilogues have no
transcendental floor.  The three resources can overlap, so max(...) is the floor of a perfectly overlapped kernel and sum(...) that of one
whose phases run one after the other."""
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'conv M=(\d+) N=(\d+) K=(\d+) k=(\d)x(\d) s=(\d) cfg=(-?\d+) ms=([\d.]+)', l)
    if m:
        rows.append(tuple(int(v) for v in m.groups()[:7]) + (float(m.group(8)),))
reid = False
print("| launch (M x N x K, kernel, stride, tile cfg) | measured ms | MFMA floor | HBM floor | SiLU floor | max | sum | measured / max |")
print("|---|---|---|---|---|---|---|---|")
tot = [0.0] * 6
for (M, N, K, kh, kw, s, cfg, ms) in rows:
    reid = reid or cfg == 101
    cin = K // (kh * kw)
    fl = 2.0 * M * N * K
    inb = M * cin * 2 * (s * s)
    outb = M * N * 2
    silu = 0 if reid else M * N
    if cfg == 102:      # front_fused: u8 frames in, layer 1 out; stem + 3x3/s2
        fl = 2.0 * M * 32 * 108 + 2.0 * (M // 4) * 64 * 288; inb = M * 4 * 3; outb = (M // 4) * 64 * 2; silu = M * 32 + (M // 4) * 64
    elif cfg == 103:    # c3_fused: the whole first C3 on 64 channels
        fl = 2.0 * M * (64 * 64 + 32 * 32 + 288 * 32 + 64 * 64); inb = M * 64 * 2; outb = M * 64 * 2; silu = M * (64 + 32 + 32 + 64)
    elif cfg == 104:    # bneck_fused: 1x1 + 3x3 on 64 channels
        fl = 2.0 * M * (64 * 64 + 576 * 64); inb = M * 64 * 2; outb = M * 64 * 2; silu = M * 128
    elif cfg == 105:    # + cv3 (128 -> 128)
        fl = 2.0 * M * (64 * 64 + 576 * 64 + 128 * 128); inb = M * 128 * 2; outb = M * 128 * 2; silu = M * 256
    elif cfg == 106:    # reid_block_fused: two 3x3 on 64 channels
        fl = 2.0 * M * 2 * 576 * 64; inb = M * 64 * 2 * 2; outb = M * 64 * 2
    elif cfg == 107:    # 3x3 / s2 (64 -> 128) + the pointwise conv that alone reads it (128 -> 128): one launch, the 3x3's output stays on chip
        fl = 2.0 * M * 128 * (576 + 128); inb = M * 64 * 2 * 4; outb = M * 128 * 2; silu = M * 256
    elif cfg == 101:    # reid stem + pool: crops in, pooled out
        inb = M * 8 * 2; outb = (M // 4) * 64 * 2
    elif N == 8 or cfg == -1:
        silu = 0        # Detect head: no activation
    wb = N * K * 2
    t_m = fl / 2.5e15 * 1e3
    t_b = (inb + wb + outb) / 6.3e12 * 1e3
    t_s = silu * 27.0 / 64.0 / 1024 / 2.3e9 * 1e3
    mx, sm = max(t_m, t_b, t_s), t_m + t_b + t_s
    for i, v in enumerate((ms, t_m, t_b, t_s, mx, sm)):
        tot[i] += v
    print(f"| {'ReID ' if reid else ''}{M} x {N} x {K}, {kh}x{kw}, s{s}, cfg {cfg} | {ms:.4f} | {t_m:.4f} | {t_b:.4f} | {t_s:.4f} | {mx:.4f} | {sm:.4f} | {ms / mx:.2f} |")
nfr = max((r[0] for r in rows if r[3] == 6), default=0) // (320 * 320)      # the stem's output pixels give the batch (640 x 640 frames)
print(f"| **all {len(rows)} launches of a {nfr if nfr else 'one'}-frame step** | **{tot[0]:.3f}** | {tot[1]:.3f} | {tot[2]:.3f} | {tot[3]:.3f} | {tot[4]:.3f} | {tot[5]:.3f} | {tot[0] / tot[4]:.2f} |")
