"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of the same bench command) into per-kernel HBM bytes per
launch.  usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md §HBM), so it
is doubled; WRITE_SIZE is taken as is (uncalibrated per the guide)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

BENCH_NAMES = [
    (r"attn_fwd32_kernel<128", "attention32_hd128"),
    (r"attn_fwd32_kernel<80", "attention32_hd80"),
    (r"attn_fwd_kernel<128, 4, false>", "attention_hd128"),
    (r"attn_fwd_kernel<80, 4, false>", "attention_hd80"),
    (r"hfre_pool_items_kernel", "hfre_pool_items"),
    (r"hfre_weights_kernel", "hfre_weights"),
    (r"hfre_finish2_kernel", "hfre_finish2"),
    (r"hfre_pool_kernel", "hfre_pool"),
    (r"dwconv3x3_ln", "dwconv3x3_ln"),
    (r"rownorm_kernel<0,", "rmsnorm"),
    (r"rownorm_kernel<1,", "layernorm"),
]


def bench_name(k):
    """rocprofv3 kernel name -> the row name fo1_profile_read / bench.py uses."""
    if "gemm_bt_p4_kernel" in k:
        return "gemm_bt_p4<256,256>"
    if "gemm_bt_p8_kernel" in k:
        return "gemm_bt_p8<256,256>"
    m = re.search(r"gemm_bt_glds_kernel<(\d+), (\d+)", k)
    if m:
        return f"gemm_bt_glds<{m.group(1)},{m.group(2)}>"
    m = re.search(r"gemm_bt_ring_kernel<(\d+), (\d+), (\d+)>", k)
    if m:
        return f"gemm_bt_ring<{m.group(1)},{m.group(2)},{m.group(3)}>"
    return next((b for pat, b in BENCH_NAMES if pat in k), k[:80])


def fold(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch = fold(sys.argv[1], "FETCH_SIZE")
    write = fold(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch (mean over all launches of the kernel in the run)",
           "corrections": "FETCH_SIZE KB x 1024 x 2 (gfx950 half-count of 16 B/lane reads); WRITE_SIZE KB x 1024",
           "command": "python bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline (default workload: 25 images x 100 boxes per pass; one rocprofv3 pass per counter)", "kernels": {}}
    for k, (n, v) in fetch.items():
        if not re.search(r"fo1::", k):
            continue
        name = bench_name(k)
        fb = v / n * 1024 * 2
        wn, wv = write.get(k, (0, 0.0))
        wb = wv / wn * 1024 if wn else 0.0
        out["kernels"][name] = dict(launches_sampled=n, fetch_bytes=round(fb), write_bytes=round(wb), hbm_bytes_per_launch=round(fb + wb))
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:12]:
        print(f"{k:40s} n={v['launches_sampled']:5d} fetch {v['fetch_bytes']/1e6:9.2f} MB write {v['write_bytes']/1e6:8.2f} MB")


if __name__ == "__main__":
    main()
