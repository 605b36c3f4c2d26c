"""Are the first tokens / logits of ONE packed pass the same on the base engine and on a replica, eagerly and under hipGraph replay?
(round 5: the 2-rank `scale` block found 32 of 64 sample items with another first token than one rank alone — all of them in the group the
second worker / replica ran.)   python scripts/replica_determinism.py [images]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    cases = [bench.build_workload(dev, n_boxes=100, img_hw=(480, 640), seed=1234 + i) for i in range(B)]
    pipe = bench.Pipeline(cases[0], dev, inflight=2, batch=B, cases=cases)
    runs = {}
    for slot in (0, 1):
        for tag, graph in (("eager", False), ("graph_seen", True), ("graph_capture", True), ("graph_replay", True), ("graph_replay2", True)):
            outs = pipe.step(graph, slot)
            torch.cuda.synchronize()
            toks = torch.cat([o["next_token"].view(-1) for o in outs]).cpu().tolist()
            lg = torch.cat([o["logits"].float().view(1, -1) for o in outs], 0)
            runs[f"slot{slot}_{tag}"] = dict(tokens=toks, logit_sum=float(lg.double().sum()), logit_absmax=float(lg.abs().max()),
                                             hidden_sum=float(torch.cat([o["last_hidden"].float() for o in outs], 0).double().sum()))
    ref = runs["slot0_eager"]
    rows = {k: dict(tokens_equal=v["tokens"] == ref["tokens"], n_diff=sum(a != b for a, b in zip(v["tokens"], ref["tokens"])),
                    logit_sum=v["logit_sum"], hidden_sum=v["hidden_sum"]) for k, v in runs.items()}
    print(json.dumps(dict(images=B, fused_qkv=os.environ.get("FO1_QKV_FUSED", "1"), attn32=os.environ.get("FO1_ATTN32", "1"), rows=rows), indent=1))


if __name__ == "__main__":
    main()
