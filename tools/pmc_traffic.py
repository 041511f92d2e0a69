"""Per-launch HBM traffic of the conv kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).

    python tools/pmc_traffic.py fetch.db write.db > profiles/rNN_pmc_traffic.json

MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
read -> doubled here; WRITE_SIZE is uncalibrated and taken as is.
"""
import json, sqlite3, sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name=? group by kernel_name", (counter,))
    return {r[0]: (r[1], r[2]) for r in rows}


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch", "fetch_correction": 2.0, "kernels": {}}
tot_b = tot_n = 0
for k in f:
    if not any(t in k for t in ("conv_igemm_kernel", "conv3x3_halo_kernel", "conv3x3_halo_v2_kernel", "conv3x3s2_halo_kernel", "conv1x1_direct_kernel", "stem_direct_kernel", "front_fused_kernel", "c3_fused_kernel", "bneck_fused_kernel", "reid_block_fused_kernel", "reid_stem_pool_kernel")):
        continue
    fb = f[k][0] * 1024 * 2.0
    wb = w.get(k, (0, 1))[0] * 1024
    n = f[k][1]
    out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / max(w.get(k, (0, 1))[1], 1)}
    tot_b += fb + wb
    tot_n += n
out["conv_all"] = {"launches": tot_n, "hbm_bytes_per_launch": tot_b / max(tot_n, 1)}
print(json.dumps(out, indent=1))
