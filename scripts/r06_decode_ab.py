"""(round 6: + the keys-per-split of the <= 32-sequence decode attention: FO1_DECODE_CHUNKS="64:2048 256:2048 1024:1024" = chunk:kv-bucket pairs)
Decode-step A/B on the GPU box: per-kernel time (library event pairs, eager) and graph-replay time per step for every combination of
the GEMV implementation (1 = MFMA skinny GEMM, 3 = the same without the 8-row units at M <= 8, 0 = v_dot2) and the decode-attention implementation (1 = workgroup per head, 0 = 64-key
split + combine), at several batch sizes, one model build.
usage: decode_ab.py [out.json] [B ...]      (default B = 1 8 16)"""
import json
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from vlm_fo1_amd import lib as L

out_path = sys.argv[1] if len(sys.argv) > 1 else None
batches = [int(a) for a in sys.argv[2:]] or [1, 8, 16]
dev = torch.device("cuda", 0)
Bmax = max(batches)
cases = [bench.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(Bmax)]
pipe = bench.Pipeline(cases[0], dev, inflight=1, batch=Bmax, cases=cases)
eng = pipe.eng
lib = L.load()
results = []
for B in batches:
    reqs = pipe.requests[:B]
    for chunk, bucket in [tuple(int(v) for v in t.split(":")) for t in os.environ.get("FO1_DECODE_CHUNKS", "64:2048").split()]:
        gemv_impl, attn_impl = 1, 0
        L.check(lib.fo1_attention_decode_set_small_chunk(chunk), "chunk")
        if os.environ.get("FO1_ATTN_TILES"):          # 64-key tiles per attention item at 17..32 sequences (the partials do not depend on it)
            L.check(lib.fo1_attention_decode_set_small_chunk(int(os.environ["FO1_ATTN_TILES"])), "tiles")
        type(eng._decoder()).KV_BUCKET = bucket
        L.check(lib.fo1_gemv_batch_set_impl(gemv_impl), "gemv impl")
        L.check(lib.fo1_attention_decode_set_impl(attn_impl), "attn impl")
        eng.prefill_batch(reqs, use_graph=False)
        d = eng._decoder()
        d._graphs = {}
        hp = eng._last_batch
        d.start(hp["seqs"], hp["delta"], eng._last_next_tokens[:B], 4096, ())
        for _ in range(3):
            d.step(False)
        torch.cuda.synchronize()
        lib.fo1_gemm_profile_shapes(1)
        L.profile(True)
        for _ in range(5):
            d.step(False)
        torch.cuda.synchronize()
        rows = L.profile_rows(reset=True)
        L.profile(False)
        lib.fo1_gemm_profile_shapes(0)
        tot = sum(r["total_ms"] for r in rows) / 5
        for _ in range(3):
            d.step(True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(32):
            d.step(True)
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t) / 32 * 1e3
        ids = d.ids[:B, :8].cpu().tolist()
        print(f"\n== B={B} chunk={chunk} bucket={bucket}: kernel time {tot:.3f} ms/step, graph replay {replay:.3f} ms/step "
              f"({B / replay * 1e3:.0f} tok/s), kv bucket {d.kv_bucket()}, slot {d.slot}")
        krows = []
        for r in sorted(rows, key=lambda r: -r["total_ms"]):
            per = r["total_ms"] / r["calls"] * 1e3
            bw = r["total_work"] / r["calls"] / (per * 1e-6) / 1e9 if per > 0 else 0
            print(f"   {r['name']:36s} x{r['calls'] // 5:4d}  {per:8.2f} us  {r['total_ms'] / 5:8.3f} ms/step  {bw:8.1f} GB/s(work)")
            krows.append(dict(name=r["name"], launches=r["calls"] // 5, us=round(per, 2), ms_per_step=round(r["total_ms"] / 5, 4), gbps_work=round(bw, 1)))
        results.append(dict(B=B, chunk=chunk, bucket=bucket, gemv_impl=gemv_impl, attn_impl=attn_impl, kernel_ms_per_step=round(tot, 4), graph_replay_ms_per_step=round(replay, 4),
                            tokens_per_sec=round(B / replay * 1e3, 1), first_ids=ids, kernels=krows))
L.check(lib.fo1_gemv_batch_set_impl(1), "gemv impl")
L.check(lib.fo1_attention_decode_set_impl(0), "attn impl")
L.check(lib.fo1_attention_decode_set_small_chunk(64), "chunk")
# first generated ids per configuration (same prefill): the implementations should agree except at near-ties
for B in batches:
    rs = [r for r in results if r["B"] == B]
    same = all(r["first_ids"] == rs[0]["first_ids"] for r in rs)
    print(f"B={B}: first 8 ids identical across implementations: {same}")
if out_path:
    with open(out_path, "w") as f:
        json.dump(results, f, indent=1)
