"""CPU tests of the N>1 path (world_size 2, gloo): cost-balanced sharding with no data-path collective and
one gather at the reducer; rank 0's merged result must equal the single-process result, including a failed
item and ragged token counts."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from vlm_fo1_amd import sharded_eval as SE


def fake_generate(i):
    if i == 5:
        raise RuntimeError("boom")
    return [(i * 7 + k) % 1000 for k in range(1 + i % 4)]


def _worker(rank, world, port, n, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    SE.init_distributed("gloo")
    costs = [(i % 5) + 1 for i in range(n)]
    merged = SE.run_sharded(n, costs, fake_generate, device="cpu")
    if rank == 0:
        q.put(merged)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_assign_is_balanced_and_complete():
    costs = [100, 1, 1, 1, 50, 50, 2, 3, 99, 4]
    for w in (1, 2, 4, 8):
        shards = SE.assign(costs, w)
        assert sorted(i for s in shards for i in s) == list(range(len(costs)))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs)
    assert SE.assign(costs, 2) == SE.assign(costs, 2)  # deterministic


def test_world_size_2_gather_equals_single_process():
    n = 23
    single = sorted([(i, None if i == 5 else fake_generate(i)) for i in range(n)], key=lambda r: r[0])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert merged == single


def test_eval_parsers_known_answers():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "evaluation"))
    from eval_coco import records_from_answer
    from eval_countbench import count_from_answer
    assert count_from_answer("<region3><region12> There are 7 objects, 2 big") == 7
    assert count_from_answer("none") == 0
    data = {"id": 42, "bbox_list": [[0, 0, 10, 20], [5, 5, 15, 25]], "score_list": [0.9, 0.5]}
    recs = records_from_answer("<ground>person</ground><objects><region1></objects><ground>ufo</ground><objects><region0></objects>",
                               data, {"person": 1})
    assert recs == [{"image_id": 42, "category_id": 1, "bbox": [5, 5, 10, 20], "score": 0.5}]


def test_worker_threads_give_the_single_worker_result():
    """Several requests in flight per rank (run_sharded with a list of callables): every item exactly once, failed items
    recorded, output sorted by item index — identical to the one-worker run whatever the interleaving."""
    import threading
    import time
    n = 37
    costs = [1.0] * n
    seen, lock = [], threading.Lock()

    def make(tag, delay):
        def gen(i):
            with lock:
                seen.append((tag, i))
            time.sleep(delay * ((i * 7) % 3))
            return fake_generate(i)
        return gen

    one = SE.run_sharded(n, costs, fake_generate, device="cpu")
    many = SE.run_sharded(n, costs, [make("a", 0.001), make("b", 0.002), make("c", 0.0)], device="cpu")
    assert many == one
    assert sorted(i for _, i in seen) == list(range(n)) and len({t for t, _ in seen}) > 1
    # no GPU / no replica(): request_workers degrades to one worker
    w = SE.request_workers(object(), lambda m, s: fake_generate, n=3)
    assert len(w) == 1


def oom_generate(i):
    if i == 4:
        raise MemoryError("HIP out of memory (simulated)")
    return [i]


def _worker_fatal(rank, world, port, n, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    SE.init_distributed("gloo")
    try:
        SE.run_sharded(n, [1.0] * n, oom_generate, device="cpu")
        q.put((rank, "no error"))
    except SE.RemoteRankFailed:
        q.put((rank, "remote"))
    except MemoryError:
        q.put((rank, "own"))
    torch.distributed.destroy_process_group()


def test_fatal_error_on_one_rank_stops_every_rank():
    """an out-of-memory error on one rank used to be re-raised there only, leaving the other rank blocked in the gather.
    Now the flag rides on the size exchange: the failing rank raises its own exception, the other RemoteRankFailed, both exit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fatal, args=(r, 2, port, 9, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owner = [r for r in range(2) if 4 in SE.assign([1.0] * 9, 2)[r]][0]
    assert got[owner] == "own" and got[1 - owner] == "remote"
    # single process: the fatal error surfaces as itself
    with pytest.raises(MemoryError):
        SE.run_sharded(9, [1.0] * 9, oom_generate, device="cpu")


def test_prefetcher_runs_ahead_in_order_and_keeps_per_item_errors():
    """run_sharded(prepare=...): the host side of every item is prepared ahead of its consumption on helper threads, handed to
    `generate` as its second argument, a failing prepare becomes that item's error record, and nothing is prepared twice."""
    import threading
    import time
    n = 29
    prepared, lock = [], threading.Lock()

    def prepare(i):
        time.sleep(0.002)
        with lock:
            prepared.append(i)
        if i == 7:
            raise ValueError("bad image")
        return {"i": i, "payload": [i] * 3}

    def gen(i, kw):
        assert kw["i"] == i
        return fake_generate(i) if i != 5 else (_ for _ in ()).throw(RuntimeError("boom"))

    def gen_group(idxs, kws):
        assert [k["i"] for k in kws] == list(idxs)
        if 5 in idxs:
            raise RuntimeError("boom")
        return [fake_generate(i) for i in idxs]

    want = [(i, None if i in (5, 7) else fake_generate(i)) for i in range(n)]
    assert SE.run_sharded(n, [1.0 + (i % 3) for i in range(n)], gen, prepare=prepare) == want
    assert sorted(prepared) == list(range(n))
    prepared.clear()
    got = SE.run_sharded(n, [1.0 + (i % 3) for i in range(n)], gen_group, batch=4, prepare=prepare, prefetch_depth=6)
    assert got == want
    # group retries re-prepare only the items of failed groups (their first result was consumed)
    assert set(prepared) == set(range(n))
    # two worker threads sharing one prefetcher
    got = SE.run_sharded(n, [1.0] * n, [gen, gen], prepare=prepare)
    assert got == want


def test_item_cost_orders_by_pixels_and_boxes():
    c = SE.item_cost
    assert c(640, 480, 100) > c(640, 480, 10) > c(320, 240, 10)
    assert c(4000, 3000, 10) == c(2048, 1536, 10)            # long side capped at 2048 like mm_utils.py:447-455
    assert c(640, 480, 100, aux="squash") > c(640, 480, 100)  # 768 x 768 aux image > 640 x 480
    assert 5000 < c(640, 480, 100) < 9000                     # ~7 TFLOP for the metric configuration
    assert SE.image_size("/nonexistent.jpg", [[0, 0, 99, 49]]) == (99, 49) and SE.image_size("/nonexistent.jpg") == (640, 480)
