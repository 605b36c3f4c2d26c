"""bench.py's launcher contract (no GPU): `python bench.py --gpus N` re-executes itself as one rank per GPU under
torch.distributed.run on 127.0.0.1 (the docstring's own invocation used to assert out for N > 1), and the
measurement tool imports nothing from tests/."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_self_spawn_command():
    p = _run(["--gpus", "4", "--steps", "7", "--warmup", "2"], FO1_BENCH_SPAWN_DRYRUN="1")
    assert p.returncode == 0, p.stderr[-2000:]
    cmd = json.loads(p.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_no_spawn_under_a_launcher_or_for_one_gpu():
    # under a launcher (WORLD_SIZE set) or with N = 1 the process goes straight on: on this box it stops at the GPU check
    for args, env in ((["--gpus", "2"], dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", FO1_BENCH_SPAWN_DRYRUN="1")),
                      (["--gpus", "1"], dict(FO1_BENCH_SPAWN_DRYRUN="1"))):
        p = _run(args, **env)
        assert "torch.distributed.run" not in p.stdout
        import torch
        if not torch.cuda.is_available():
            assert p.returncode != 0 and "bench.py needs a GPU" in p.stderr


def test_measurement_tools_do_not_import_tests():
    for f in ("bench.py", "__graft_entry__.py", "bench_workloads.py"):
        src = open(os.path.join(ROOT, f)).read()
        assert not re.search(r"""['"]tests['"]""", src), f"{f} puts tests/ on sys.path"
        assert "hfre_cases import" not in src.replace("fixtures.hfre_cases import", ""), f"{f} imports the test helper module"
