"""MFMA utilisation per conv instantiation = SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / (1024 SIMDs x kernel duration
x clock), from the --pmc pass (pmc_summary.py text) joined with the --kernel-trace averages (prof_summary.py table).
The PMC pass and the trace are separate runs of the same command (guide: never combine them).

    python tools/mfma_util.py gpurun_out/r01/pmc_mfma.txt gpurun_out/r01/kernel_stats.md > profiles/r01_mfma_util.md
"""
import re, sys
CLOCK_GHZ = 2.4          # peak engine clock (MI355X_MICROARCH.md); the sustained clock under load is lower, so this under-reports
busy, n = {}, {}
for l in open(sys.argv[1]):
    m = re.match(r"(.*) (SQ_\S+) (\S+) n= (\d+)$", l.strip())        # (names are cut at 70 characters by pmc_summary.py: they need not end with ")")
    if m and m.group(2) == "SQ_VALU_MFMA_BUSY_CYCLES" and any(t in m.group(1) for t in ("conv_igemm_kernel", "conv3x3_halo_kernel", "conv3x3_halo_v2_kernel", "conv3x3s2_halo_kernel", "conv1x1_direct_kernel", "stem_direct_kernel", "front_fused_kernel", "c3_fused_kernel", "bneck_fused_kernel", "reid_block_fused_kernel", "reid_stem_pool_kernel")):
        busy[m.group(1)] = float(m.group(3)); n[m.group(1)] = int(m.group(4))
dur = {}
for l in open(sys.argv[2]):
    m = re.match(r"\| `(.*)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", l)
    if m:
        dur[m.group(1)[:70]] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
print("# MFMA busy fraction of the conv kernels (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES, separate pass)\n")
print("busy = SQ_VALU_MFMA_BUSY_CYCLES per launch / 1024 SIMDs; utilisation = busy / (average kernel duration x %.1f GHz).\n" % CLOCK_GHZ)
print("| kernel | launches (pmc pass) | MFMA busy cycles / SIMD / launch | avg duration us (trace) | MFMA utilisation |")
print("|---|---|---|---|---|")
tb = td = 0.0
for k in sorted(busy, key=lambda k: -busy[k]):
    d = dur.get(k[:70])
    if not d:
        continue
    per = busy[k] / n[k] / 1024
    util = per / (d[2] * 1e-6 * CLOCK_GHZ * 1e9)
    tb += busy[k] / 1024; td += n[k] * d[2] * 1e-6 * CLOCK_GHZ * 1e9
    print(f"| `{k}` | {n[k]} | {per:,.0f} | {d[2]:.1f} | {100 * util:.1f} % |")
print(f"\nall conv launches, duration-weighted: {100 * tb / td:.1f} % MFMA busy")
if len(sys.argv) > 3:      # tools/ubench/clock_probe: the shader clock the chip holds under an MFMA + transcendental load (2.4 GHz is the peak clock)
    ghz = float(sys.argv[3])
    print(f"at the measured shader clock of {ghz:.2f} GHz: {100 * tb / td * CLOCK_GHZ / ghz:.1f} % (the same busy cycles over the cycles that actually elapsed)")
