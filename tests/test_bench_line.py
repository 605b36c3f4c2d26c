"""The stdout line of bench.py (VERDICT r5 #2b): the contract keys verbatim, at most 6 KB, the flat `summary` last — checked on the full
record of the round's campaign (profiles/r06_bench_default.json), no GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def test_stdout_line_is_compact_and_keeps_the_contract():
    import bench
    with open(os.path.join(ROOT, "profiles", "r06_bench_default.json")) as f:
        full = json.load(f)
    line = bench.compact_line(dict(full, full_record="gpurun_out/bench_full.json"))
    text = json.dumps(line)
    assert len(text) <= 6144, len(text)
    for k in CONTRACT:
        assert k in line, k
        if k not in ("roofline", "cpu_baseline"):
            assert line[k] == full[k], k
    assert list(line)[-1] == "summary"
    assert line["config"]["workload"] == full["config"]["workload"]
    roof, cpu = line["roofline"], line["cpu_baseline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_us", "algorithmic_work_per_launch"):
        assert k in roof, k
    assert abs(roof["algorithmic_work_per_launch"] / (roof["avg_us"] * 1e-6) / 1e12 - roof["achieved"]) < 0.01 * roof["achieved"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert cpu[k] == full["cpu_baseline"][k], k
    s = line["summary"]
    assert s["value_images_per_sec"] == float(f"{full['value']:.5g}")
    for k in ("end_to_end_images_per_sec", "driver_level_images_per_sec", "decode_pool_ms_per_step", "gemm_frac_of_peak", "hfre_hbm_frac"):
        assert k in s, k
    # the nested numbers a reader looks up by path are still where the full record has them
    assert line["decode"]["pool"]["ms_per_step"] == float(f"{full['decode']['pool']['ms_per_step']:.6g}")
    assert line["end_to_end"]["images_per_sec"] == float(f"{full['end_to_end']['images_per_sec']:.6g}")
