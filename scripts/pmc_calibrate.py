#!/usr/bin/env python3
"""HBM counter calibration (VERDICT r5 #2a).  Two roles:
  pmc_calibrate.py run                  launch every traffic-probe pattern (fo1_traffic_probe, test / bench build) on a buffer larger than the
                                        Infinity Cache — the command rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE wraps;
  pmc_calibrate.py fold <fetch_dir> <write_dir> <out.json>
                                        known bytes / counter bytes per pattern -> calibration factors.
GPU box only for `run`."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES = 1 << 30           # 1 GiB per launch: four times the 256 MB Infinity Cache
LD = 4096                 # row pitch of the tile patterns: N = 2048 bf16 columns (LLM o / down outputs, K = 2048 operands)
REPS = 3
MODES = {0: "stream_store16", 1: "tile_store", 2: "stream_load16", 3: "tile_load_lds", 4: "stream_store8", 5: "stream_store4"}


def run():
    os.environ["FO1_AB"] = "1"
    sys.path.insert(0, ROOT)
    import torch
    from vlm_fo1_amd import lib as L
    lib = L.load()
    buf = torch.zeros(BYTES, dtype=torch.uint8, device="cuda")
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(REPS):
        for m in MODES:
            L.check(lib.fo1_traffic_probe(m, buf.data_ptr(), BYTES, LD, 0, sink.data_ptr(), st), "traffic_probe")
    torch.cuda.synchronize()
    print("ran", REPS, "x", len(MODES), "patterns of", BYTES, "bytes")


def fold(fetch_dir, write_dir, out):
    def read(d, counter):
        acc = defaultdict(list)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter and "traffic_probe_kernel" in r["Kernel_Name"]:
                    acc[int(r.get("Dispatch_Id", 0))].append(float(r["Counter_Value"]))
        return [sum(v) for _, v in sorted(acc.items())]      # dispatch order = REPS x modes in MODES order
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    res = {"bytes_per_launch": BYTES, "row_pitch_bytes_of_tile_patterns": LD, "counter_unit": "KB as rocprofv3 reports it (x 1024 below)",
           "note": "factor = known bytes / (counter x 1024): multiply a kernel's counter reading by the factor of ITS access pattern; a load pattern's "
                   "WRITE_SIZE and a store pattern's FETCH_SIZE are reported as the background the counter sees beside it", "patterns": {}}
    nm = len(MODES)
    for j, (m, name) in enumerate(MODES.items()):
        f = [fetch[k] for k in range(j, len(fetch), nm)]
        w = [write[k] for k in range(j, len(write), nm)]
        fb = sum(f) / max(1, len(f)) * 1024
        wb = sum(w) / max(1, len(w)) * 1024
        is_store = "store" in name
        res["patterns"][name] = dict(known_bytes=BYTES, fetch_counter_bytes=round(fb), write_counter_bytes=round(wb),
                                     factor=round(BYTES / (wb if is_store else fb), 4) if (wb if is_store else fb) > 0 else None,
                                     calibrates="WRITE_SIZE" if is_store else "FETCH_SIZE", launches=len(w if is_store else f))
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["patterns"].items():
        print(f"{k:20s} {v['calibrates']:10s} factor {v['factor']}  (fetch {v['fetch_counter_bytes'] / 1e6:.1f} MB, write {v['write_counter_bytes'] / 1e6:.1f} MB of {BYTES / 1e6:.1f} MB)")


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) == 5 and sys.argv[1] == "fold":
        fold(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
