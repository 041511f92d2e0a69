"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) into a markdown table.

    python tools/prof_summary.py gpurun_out/prof_r1/r1_results.db "command line" > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(db, cmd):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                          "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats\n\ncommand: `{cmd}`\n\nGPU kernel time total: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx in rows:
        print(f"| `{name}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    conv = [r for r in rows if any(t in r[0] for t in ("conv_igemm_kernel", "conv3x3_halo_kernel", "conv3x3_halo_v2_kernel", "conv3x3s2_halo_kernel", "conv1x1_direct_kernel", "stem_direct_kernel", "front_fused_kernel", "c3_fused_kernel", "bneck_fused_kernel", "reid_block_fused_kernel", "reid_stem_pool_kernel"))]
    if conv:
        n, tot = sum(r[1] for r in conv), sum(r[2] for r in conv)
        print(f"\nall conv kernels (`conv_igemm_kernel<*>`, `conv3x3_halo_kernel<*>`, `conv3x3_halo_v2_kernel<*>`, `conv3x3s2_halo_kernel<*>`, `conv1x1_direct_kernel<*>`, `stem_direct_kernel<*>`, `front_fused_kernel<*>`, `c3_fused_kernel`, `bneck_fused_kernel`, `reid_block_fused_kernel`, `reid_stem_pool_kernel`): {n} launches, {tot / 1e6:.3f} ms, average {tot / n / 1e3:.2f} us per launch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
