"""Compare bench.py's in-process per-kernel averages (hipExtLaunchKernelGGL timestamps) with rocprofv3's kernel-trace averages
of the same command.  usage: check_profile_agreement.py <bench.json> <kernel_stats.csv>"""
import csv
import json
import re
import sys

b = json.load(open(sys.argv[1]))["roofline"]
stats = {}
for r in csv.DictReader(open(sys.argv[2])):
    stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)


def rocprof_avg(bench_name):
    pats = {"attn_fwd": r"attn_fwd_kernel<\d+, 4, false>", "attn_fwd32": r"attn_fwd32_kernel<", "rmsnorm": r"rownorm_kernel<0[,>]", "layernorm": r"rownorm_kernel<1[,>]",
            "dwconv3x3_ln": r"dwconv3x3_ln(_run)?_kernel<", "chattn_gram": r"chattn_gram(_mfma)?_kernel", "chattn_apply": r"chattn_apply(_mfma)?_kernel",
            "win_attn32": r"win_attn32_kernel<"}
    m = re.match(r"gemm_bt_p(\d)<", bench_name)
    if m:
        sel = [(c, a) for k, (c, a) in stats.items() if f"gemm_bt_p{m.group(1)}_kernel" in k]
        n = sum(c for c, _ in sel)
        return sum(c * a for c, a in sel) / n if n else None
    m = re.match(r"gemm_bt_glds<(\d+),(\d+)>", bench_name)
    if m:
        pat = rf"gemm_bt_glds_kernel<{m.group(1)}, {m.group(2)},"
    else:
        m = re.match(r"gemm_bt_ring<(\d+),(\d+),(\d+)>", bench_name)
        pat = rf"gemm_bt_ring_kernel<{m.group(1)}, {m.group(2)}, {m.group(3)}>" if m else pats.get(bench_name, re.escape(bench_name) + "_kernel")
    sel = [(c, a) for k, (c, a) in stats.items() if re.search(pat, k)]
    n = sum(c for c, _ in sel)
    return sum(c * a for c, a in sel) / n if n else None


print(f"{'kernel':28s} {'bench us':>9s} {'rocprof us':>10s} {'ratio':>6s}")
for name, ms in list(b["per_step_ms"].items())[:14]:
    avg = ms / b["launches"][name] * 1e3
    ra = rocprof_avg(name)
    print(f"{name:28s} {avg:9.1f} {ra if ra is None else round(ra, 1)!s:>10s} {'' if not ra else f'{avg / ra:6.2f}'}")
