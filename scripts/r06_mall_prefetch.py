#!/usr/bin/env python3
"""Round 6 experiment: does a weight PREFETCH into the Infinity Cache (a streaming read of the next projection's weights, concurrent with the
current kernel) shorten the decode step's weight-streaming kernels?  For the single-sequence GEMVs (gate/up, down, q/k/v, o) and the pool's
gate/up GEMM: (cold) weights rotate over > 1 GB of copies; (serial) stream-read W_j, then the kernel on W_j — kernel time from a warm Infinity
Cache = serial - prefetch alone; (overlap) the stream-read of W_{j+1} runs on a second stream beside the kernel on W_j.  All as hipGraph replays.
GPU box only.  usage: r06_mall_prefetch.py <out.json>"""
import os
os.environ.setdefault("FO1_AB", "1")
import json
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

lib = L.load()
dev = torch.device("cuda", 0)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
res = []


def prefetch(w, wgs):
    L.check(lib.fo1_traffic_probe(2, w.data_ptr(), w.numel() * 2, 0, wgs, sink.data_ptr(), torch.cuda.current_stream().cuda_stream), "probe")


def graph_time(build, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        build()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            build()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


CASES = [("gemv gate/up B=1", 1, 22016, 2048, "swiglu"), ("gemv down B=1", 1, 2048, 11008, "res"), ("gemv qkv-sized B=1", 1, 2560, 2048, "plain"),
         ("gemv gate/up B=25", 25, 22016, 2048, "swiglu"), ("gemm gate/up P=128", 128, 22016, 2048, "swiglu"), ("gemm down planes P=128", 128, 2048, 11008, "planes")]
for name, M, N, K, kind in CASES:
    x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    n = max(12, int(1.3e9 / (N * K * 2)) + 1)
    ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(n)]
    r = torch.randn(M, N, device=dev).bfloat16()
    part = torch.empty(8 * M * N, dtype=torch.float32, device=dev) if kind == "planes" else None

    def kern(w):
        if M <= 32:
            if kind == "swiglu":
                return ops.gemv_batch(x, w, mode=ops.GB_SWIGLU)
            return ops.gemv_batch(x, w, residual=r if kind == "res" else None)
        if kind == "planes":
            return ops.gemm_partials(x, w, 8, part)
        return ops.gemm(x, w, act=ops.ACT_SWIGLU16)

    def cold():
        for w in ws:
            kern(w)

    row = dict(case=name, weight_mb=round(N * K * 2 / 1e6, 1), copies=n)
    row["cold_us"] = round(graph_time(cold) / n, 2)
    for wgs in (256, 1024):
        def pf_only():
            for w in ws:
                prefetch(w, wgs)

        def serial():
            for w in ws:
                prefetch(w, wgs)
                kern(w)

        side = torch.cuda.Stream()

        def overlap():
            main = torch.cuda.current_stream()
            prefetch(ws[0], wgs)
            for j, w in enumerate(ws):
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)                    # the prefetch of W_{j+1} starts with the kernel on W_j
                with torch.cuda.stream(side):
                    if j + 1 < n:
                        prefetch(ws[j + 1], wgs)
                    ev2 = torch.cuda.Event()
                    ev2.record(side)
                kern(w)
                main.wait_event(ev2)                   # the next kernel starts when both are done

        pf = graph_time(pf_only) / n
        se = graph_time(serial) / n
        ov = graph_time(overlap) / n
        row[f"wg{wgs}"] = dict(prefetch_alone_us=round(pf, 2), prefetch_tbps=round(N * K * 2 / pf / 1e6, 2), serial_us=round(se, 2),
                               kernel_from_warm_cache_us=round(se - pf, 2), overlapped_us=round(ov, 2))
    print(json.dumps(row), flush=True)
    res.append(row)
    del ws
    torch.cuda.empty_cache()
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_mall_prefetch.json", "w"), indent=1)
