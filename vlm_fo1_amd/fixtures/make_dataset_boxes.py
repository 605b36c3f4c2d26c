"""Extracts vlm_fo1_amd/fixtures/dataset_boxes.npz: the UPN box lists (data, not code) of EVERY item of the reference's two evaluation fixtures
(/root/reference/evaluation/processed_data/{countbench,pixmoCount}_with_upn_score_0.3_0.8.json, SURVEY §8c/§8d cfg4: 487 items /
11 144 boxes and 529 items / 28 996 boxes, integer pixels, N in [2, 100]) as int16 arrays, so that bench.py's dataset-shaped workloads
and the GPU-box tests have the real variable-N box geometry without reading /root/reference at run time.  Images are not part of the
fixtures (none exist offline): workloads synthesise each image at the extent of its boxes, max(x2) x max(y2).

    python vlm_fo1_amd/fixtures/make_dataset_boxes.py
"""
import json
import os

import numpy as np

SRC = "/root/reference/evaluation/processed_data/"
FILES = (("countbench", "countbench_with_upn_score_0.3_0.8.json"), ("pixmo", "pixmoCount_with_upn_score_0.3_0.8.json"))


def main():
    out = {}
    for name, f in FILES:
        d = json.load(open(SRC + f))
        counts = np.array([len(x["bboxes"]) for x in d], dtype=np.int16)
        boxes = np.array([b for x in d for b in x["bboxes"]], dtype=np.int64)
        assert boxes.min() >= -32768 and boxes.max() < 32768 and np.all(boxes == np.round(boxes))
        out[name + "_counts"] = counts
        out[name + "_boxes"] = boxes.astype(np.int16)
        print(name, len(d), "items", boxes.shape[0], "boxes", "N", counts.min(), "..", counts.max())
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataset_boxes.npz"), **out)


if __name__ == "__main__":
    main()
