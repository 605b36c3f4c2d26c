"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of the same bench command) into per-kernel HBM bytes per
launch.  usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md §HBM), so it
is doubled; WRITE_SIZE is taken as is.  Both factors are CALIBRATED on this engine's own access patterns (round 6, profiles/r06_pmc_calibration.json:
scripts/pmc_calibrate.py moves a known 1 GiB per pattern — 16 / 8 / 4-byte streaming stores and the 256 x 256 GEMM's coalesced tile store read
1.0000 x on WRITE_SIZE, streaming loads and the GEMM's LDS-DMA tile loads 2.000 x on FETCH_SIZE).
Round 6 fix: every template instantiation of a kernel is its own rocprofv3 kernel name; rounds 3-5 wrote the per-kernel row once per
instantiation under the SAME bench name, so the row held whichever instantiation came last (r05: the 36 fused-q/k/v launches of the LLM, whose
83 MB of writes were then read against the 176 MB mean output of all 384 launches).  Instantiations are now summed, launch-weighted.
An optional 4th argument names a `bench.py --profile-shapes` record: its per-shape rows give the ALGORITHMIC read / write bytes per launch that
the row of the 256 x 256 GEMM is set against (reads: A + W (+ residual operand) once; writes: C once)."""
import csv
import glob
import json
import os
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


def algorithmic_gemm_bytes(per_shape_json):
    """Launch-weighted mean over one packed pass of what a launch of the 256 x 256 GEMM must read (A once — an implicit convolution reads its padded
    map once: M x Cin —, W once, the residual operand of o / down / proj / fc2 products once, rope tables of the fused q/k/v form) and write (C once; N / 2
    columns behind a SwiGLU epilogue; the fused q/k/v form writes q (+ k in the ViT layout) and the K / V^T cache rows)."""
    d = json.load(open(per_shape_json))
    d = d.get("all_on", d)
    d = d.get("roofline", d)
    rd = wr = n = 0
    for k, _ in d["per_step_ms"].items():
        m = re.match(r"gemm (\d+)x(\d+)x(\d+) t256x256 (\w+)", k)
        if not m:
            continue
        M, N, K = map(int, m.groups()[:3])
        kind, c = m.group(4), d["launches"][k]
        a = M * (K // 9 if kind == "conv" else K) * 2
        w = N * K * 2
        if kind == "qkv0":          # LLM: q columns out, k / v to the caches, bf16 cos / sin rows in
            out, extra = M * N * 2, M * 128 * 2 * 2
        elif kind == "qkv1":        # ViT head-major: [q 80 | k 80 | - | -] of every 256 out, V^T 80 of every 256, fp32 cos / sin [M, 40] in
            out, extra = M * N * 2 * 240 // 256, M * 40 * 4 * 2
        else:
            swiglu = N in (22016, 6912)
            out = M * N * (1 if swiglu else 2)
            # residual epilogues: the LLM's o / down (N = 2048), the ViT's proj / down (N = 1280) and DaViT's proj / fc2 products (N = C of the stage, K in {C, 4C})
            has_res = (N in (2048, 1280) and K in (2048, 11008, 1280, 3456)) or (N in (256, 512, 1024, 2048) and K in (N, 4 * N) and kind == "s1" and M >= 20000)
            extra = M * N * 2 if has_res else 0
        rd += (a + w + extra) * c
        wr += out * c
        n += c
    return dict(read=rd / n, write=wr / n, launches=n) if n else None


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
    agg = defaultdict(lambda: [0, 0.0, 0, 0.0, 0])          # bench name -> [fetch launches, fetch KB, write launches, write KB, instantiations]
    for k, (n, v) in fetch.items():
        if not re.search(r"fo1::", k):
            continue
        a = agg[bench_name(k)]
        a[0] += n
        a[1] += v
        wn, wv = write.get(k, (0, 0.0))
        a[2] += wn
        a[3] += wv
        a[4] += 1
    for name, (n, v, wn, wv, ni) in agg.items():
        fb = v / n * 1024 * 2
        wb = wv / wn * 1024 if wn else 0.0
        out["kernels"][name] = dict(launches_sampled=n, instantiations=ni, fetch_bytes=round(fb), write_bytes=round(wb), hbm_bytes_per_launch=round(fb + wb))
    if len(sys.argv) > 4:
        alg = algorithmic_gemm_bytes(sys.argv[4])
        row = out["kernels"].get("gemm_bt_p4<256,256>")
        if alg and row:
            row.update(algorithmic_read_bytes=round(alg["read"]), algorithmic_write_bytes=round(alg["write"]), algorithmic_launches_per_pass=alg["launches"],
                       read_over_algorithmic=round(row["fetch_bytes"] / alg["read"], 3), write_over_algorithmic=round(row["write_bytes"] / alg["write"], 3),
                       algorithmic_source=os.path.basename(sys.argv[4]))
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:12]:
        print(f"{k:40s} n={v['launches_sampled']:5d} fetch {v['fetch_bytes']/1e6:9.2f} MB write {v['write_bytes']/1e6:8.2f} MB")


if __name__ == "__main__":
    main()
