"""fp8 linears inside the engine (FO1Engine.enable_fp8, BASELINE configs[4]) against the bf16 engine on the metric's configuration at
FULL depth: the deviation table of DESIGN.md section 10 comes from this test (gpurun_out/fp8_engine_metrics.json).  The reference has
no fp8 path; the bars below are the first measurement with a 1.5x margin, and the bf16 engine itself is pinned to the oracle by
tests/test_fulldepth_parity_gpu.py."""
import json
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _m(a, b):
    a, b = a.float().reshape(-1, a.shape[-1]), b.float().reshape(-1, b.shape[-1])
    cos = F.cosine_similarity(a, b, dim=-1)
    return dict(min_cos=float(cos.min()), mean_cos=float(cos.mean()), rel=float((a - b).abs().max() / b.abs().max()),
                rms_rel=float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()))


@pytest.mark.parametrize("preset", ["all", "mlp"])
def test_fp8_engine_tracks_bf16_engine_full_depth(preset):
    import bench
    from vlm_fo1_amd import ops
    dev = torch.device("cuda", 0)
    B = 2
    cases = [bench.build_workload(dev, n_boxes=100, seed=500 + i) for i in range(B)]
    pipe = bench.Pipeline(cases[0], dev, inflight=1, batch=B, cases=cases)
    eng = pipe.eng
    KEYS = ("image_tokens", "region_tokens", "last_hidden", "logits")

    def run(graph):
        outs = eng.prefill_batch(pipe.requests, use_graph=graph)
        return {k: torch.cat([o[k] for o in outs]).clone() for k in KEYS}

    ref = run(False)
    try:
        n = eng.enable_fp8(preset)
        assert n == (32 * 3 + 36 * 3 if preset == "all" else 32 * 2 + 36 * 2)
        got = run(False)
        # graph replay of the fp8 pass == eager fp8 pass, bit for bit
        for _ in range(3):
            g = run(True)
        assert torch.equal(g["logits"], got["logits"]) and torch.equal(g["region_tokens"], got["region_tokens"])
    finally:
        eng.disable_fp8()
    again = run(False)
    assert torch.equal(again["logits"], ref["logits"]), "disable_fp8 must restore the bf16 path exactly"
    M = {k: _m(got[k], ref[k]) for k in KEYS}
    lg, lr = got["logits"].float(), ref["logits"].float()
    top2 = lr.topk(2, dim=-1).values
    M["logits"].update(max_abs=float((lg - lr).abs().max()), ref_std=float(lr.std()), min_margin=float((top2[:, 0] - top2[:, 1]).min()),
                       argmax_equal=bool(torch.equal(lg.argmax(-1), lr.argmax(-1))))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(M, open(os.path.join(ROOT, "gpurun_out", f"fp8_engine_metrics_{preset}.json"), "w"), indent=1)
    print(preset, json.dumps(M, indent=1))
    # first measurement (random gaussian weights, which neither damp nor learn around quantisation noise: every e4m3 product carries
    # ~3 % rms error and 32 + 36 residual blocks accumulate it): "all" image tokens 0.983 / region 0.9995 / hidden 0.933 / logits 0.933
    assert M["image_tokens"]["min_cos"] >= 0.97 and M["region_tokens"]["min_cos"] >= 0.999
    assert M["last_hidden"]["min_cos"] >= 0.90 and M["logits"]["min_cos"] >= 0.90
