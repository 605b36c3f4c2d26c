"""Extracts vlm_fo1_amd/fixtures/box_fixtures.json — a SMALL sample of the real UPN box lists
(data, not code) from the reference's evaluation fixtures
(/root/reference/evaluation/processed_data/*.json) so that GPU-box tests and
bench.py have real box geometry without reading /root/reference at run time."""
import json

SRC = "/root/reference/evaluation/processed_data/"
FILES = (("countbench", "countbench_with_upn_score_0.3_0.8.json"), ("pixmo", "pixmoCount_with_upn_score_0.3_0.8.json"))


def main():
    out = {}
    for name, f in FILES:
        d = json.load(open(SRC + f))
        picks = []
        want = [lambda n: n == 2, lambda n: 30 <= n <= 40, lambda n: n == 100, lambda n: 5 <= n <= 9,
                lambda n: 60 <= n <= 80]
        for w in want:
            for i, x in enumerate(d):
                if w(len(x["bboxes"])) and i not in picks:
                    picks.append(i)
                    break
        for i in range(3):
            if i not in picks:
                picks.append(i)
        items = []
        for i in sorted(picks):
            x = d[i]
            xs = max(b[2] for b in x["bboxes"])
            ys = max(b[3] for b in x["bboxes"])
            items.append({"index": i, "question": x.get("question"), "answer": x.get("answer"),
                          "bboxes": x["bboxes"], "extent": [xs, ys]})
        out[name] = items
        print(name, [(it["index"], len(it["bboxes"]), it["extent"]) for it in items])
    json.dump(out, open("vlm_fo1_amd/fixtures/box_fixtures.json", "w"))


if __name__ == "__main__":
    main()
