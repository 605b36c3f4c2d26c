"""N > 1 on real devices (RCCL): run only when the box exposes >= 2 GPUs (the 1-GPU test box skips them; the driver's 8-GPU scaling
run and any multi-GPU box execute them).  (a) bench.py --gpus 2 under torch.distributed.run over the nccl backend: one JSON line from
rank 0, weak scaling, value > 0;  (b) sharded_eval.run_sharded over nccl: the one all_gather of device tensors merges the two ranks'
records exactly like the single-process run (reference eval loops evaluation/eval_coco.py:36, eval_countbench.py:22)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
need2 = pytest.mark.skipif(_NDEV < 2, reason=f"RCCL tests skipped: this box exposes {_NDEV} GPU(s), they need >= 2 (RCCL refuses two ranks on one "
                           "device; the gloo world-2 tests in tests/test_sharded_eval.py cover the same control flow on CPU)")

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch
from vlm_fo1_amd import sharded_eval as SE
rank, world, local = SE.init_distributed("nccl")
def gen(i):
    if i == 5:
        raise RuntimeError("boom")
    return [(i * 7 + k) % 1000 for k in range(1 + i % 4)]
n = 23
merged = SE.run_sharded(n, [(i % 5) + 1 for i in range(n)], gen, device=f"cuda:{local}")
if rank == 0:
    json.dump(merged, open(sys.argv[2], "w"))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def _torchrun(args, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)


@need2
def test_bench_two_gpus_over_rccl():
    p = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], 29541)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["global_batch"] == 2 * out["config"]["images_per_step"]
    # the nccl twin of tests/test_e2e_gpu.py::test_bench_two_ranks_control_flow's `scale` checks: the sharded evaluation loop across two
    # real devices, the one all_gather over RCCL, merged ids of the sample == one rank alone
    sc = out["scale"]
    assert sc["world_size"] == 2 and sc["dist_world_size"] == 2 and sc["backend"] == "nccl" and sc["one_device_gloo_test_mode"] is False
    assert sc["per_rank_items"] == [sc["items_per_gpu"]] * 2 and len(sc["gather_ms"]) == 2 and sc["images_per_sec"] > 0
    assert sc["items_failed"] == 0 and sc["sample_ids_equal_to_one_rank_alone"] is True, sc


@need2
def test_run_sharded_over_rccl(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    out = tmp_path / "merged.json"
    p = _torchrun([str(script), ROOT, str(out)], 29542)
    assert p.returncode == 0, p.stderr[-3000:]
    merged = [tuple(r) for r in json.load(open(out))]
    single = [(i, None if i == 5 else [(i * 7 + k) % 1000 for k in range(1 + i % 4)]) for i in range(23)]
    assert [(i, t) for i, t in merged] == single
