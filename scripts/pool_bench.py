#!/usr/bin/env python3
"""Decode-pool step timing on the GPU: ms per step and per-kernel times of llm.DecodePool at 64 / 128 slots, against the <= 32-sequence BatchDecoder, at the metric configuration's
context (651-token prompts).  Prints one JSON object; `python scripts/pool_bench.py [--slots 64 128] [--chunks 64 128 256]`."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, nargs="+", default=[64, 128])
    ap.add_argument("--chunks", type=int, nargs="+", default=[0], help="keys per split of the pool's decode attention to A/B (needs FO1_AB=1; 0 = default)")
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--attn-impl", type=int, nargs="+", default=[0], help="A/B (FO1_AB=1): 0 = split-KV + combine, 1 = one workgroup per (KV head, sequence)")
    ap.add_argument("--splits", type=int, nargs=4, default=None, metavar=("QKV", "O", "DOWN", "GATEUP"), help="split-K planes of the pool's q/k/v, o, down and gate/up projections (default: DecodePool.SPLITS; GATEUP 0 = SwiGLU epilogue)")
    ap.add_argument("--tiled", action="store_true", help="gate/up and lm_head from the pre-tiled weight copies (DecodePool.TILED_WEIGHTS = True; default off)")
    ap.add_argument("--no-tiled", action="store_true", help="(default) gate/up and lm_head from the row-major weights")
    ap.add_argument("--unfused", action="store_true", help="the step on plain GEMM epilogues + separate RMSNorm launches (DecodePool.FUSED_SPLITK = False)")
    ap.add_argument("--fill", type=float, default=1.0, help="fraction of the slots that hold live sequences")
    args = ap.parse_args()
    import bench as B
    from vlm_fo1_amd import lib as L
    from vlm_fo1_amd.llm import BatchDecoder, DecodePool
    if args.splits:
        DecodePool.SPLITS = dict(qkv=args.splits[0], o=args.splits[1], down=args.splits[2], gateup=args.splits[3])
    if args.unfused:
        DecodePool.FUSED_SPLITK = False
    if args.no_tiled:
        DecodePool.TILED_WEIGHTS = False
    if args.tiled:
        DecodePool.TILED_WEIGHTS = True
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cases = [B.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(32)]
    pipe = B.Pipeline(cases[0], dev, inflight=1, batch=32, cases=cases)
    eng = pipe.eng
    out = dict(tiled_weights=DecodePool.TILED_WEIGHTS, fused_splitk=DecodePool.FUSED_SPLITK, splits=dict(DecodePool.SPLITS), prompt_tokens=len(cases[0]["ids"]) - 1 + cases[0]["grid"][0] * cases[0]["grid"][1] // 4)
    wbytes = float(sum(t.numel() * t.element_size() for t in eng.llm.decode_weight_tensors()))
    out["weight_bytes_per_step"] = wbytes
    eng.prefill_batch(pipe.requests, use_graph=False)
    torch.cuda.synchronize()
    hp, first = eng._last_batch, eng._last_next_tokens.clone()

    def kv_bytes(n_seq, L):
        c = eng.llm.cfg
        return 2.0 * n_seq * c.num_layers * c.num_kv_heads * c.head_dim * 2 * L

    # ---- the <= 32-sequence decoder (round 3's path) ----
    d = BatchDecoder(eng.llm)
    d.start(hp["seqs"], hp["delta"], first, 4096, ())
    for _ in range(4):
        d.step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d.step(True)
    torch.cuda.synchronize()
    t32 = (time.perf_counter() - t0) / args.steps
    out["batch_decoder_32"] = dict(ms_per_step=round(t32 * 1e3, 3), tokens_per_sec=round(32 / t32, 1), hbm_frac=round(wbytes / t32 / 8e12, 4))

    for P in args.slots:
        for be0 in [(c, i) for c in args.chunks for i in args.attn_impl]:
            be, impl = be0
            if impl or len(args.attn_impl) > 1:
                L.check(L.load().fo1_attention_decode_set_impl(impl), "impl")
            if be:
                L.check(L.load().fo1_attention_decode_set_pool_chunk(be), "chunk")
            pool = DecodePool(eng.llm, slots=P)
            n_live = max(1, int(round(P * args.fill)))
            left = n_live
            while left > 0:
                n = min(32, left)
                pool.join(eng.llm.kcache, eng.llm.vtcache, hp["seqs"][:n], hp["delta"][:n], first[:n], 300, ())
                left -= n
            for _ in range(4):
                pool.step(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pool.step(True)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / args.steps
            L_ctx = hp["seqs"][0][1] + 4 + args.steps // 2
            row = dict(ms_per_step=round(t * 1e3, 3), live=n_live, tokens_per_sec=round(n_live / t, 1),
                       hbm_frac_weights=round(wbytes / t / 8e12, 4), hbm_frac_weights_plus_kv=round((wbytes + kv_bytes(n_live, L_ctx)) / t / 8e12, 4))
            # per-kernel times of ONE eager step (dispatch timestamps)
            L.profile(True)
            L.profile_rows(reset=True)
            pool.step(False)
            torch.cuda.synchronize()
            rows = L.profile_rows(reset=True)
            L.profile(False)
            agg = {}
            for r in rows:
                a = agg.setdefault(r["name"], [0, 0.0])
                a[0] += r["calls"]
                a[1] += r["total_ms"]
            row["kernels_ms_per_step"] = {k: [v[0], round(v[1], 4)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
            row["sum_kernel_ms"] = round(sum(v[1] for v in agg.values()), 3)
            out[f"pool_{P}" + (f"_chunk{be}" if be else "") + (f"_impl{impl}" if len(args.attn_impl) > 1 or impl else "")] = row
            del pool
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
