"""Multi-GPU evaluation: images shard embarrassingly, predictions meet once at the reducer.

The reference's eval loops are `for data in tqdm(data_list)` with no cross-item state
(evaluation/eval_coco.py:36, eval_countbench.py:22) and no distributed code at all.  Here: one process per
GPU (torchrun), a full engine replica per rank, a static cost-balanced assignment of items to ranks, NO
collective on the data path, and exactly one RCCL all_gather of fixed-width records
(item index, status, generated token ids) over xGMI at the end — a few MB for COCO-5k, latency-bound.
Rank 0 sorts by item index and runs the unchanged decode / parse / dump code, so the output is
byte-identical to a 1-GPU run.  Backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

ERR = -1  # status of an item whose generation raised (the reference silently drops it: eval_coco.py:60-65)


def world_info() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, world, local = world_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def assign(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first: sort by cost descending, give each item to the least-loaded rank.
    Deterministic (ties by index), so every rank computes the same assignment with no communication."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    for s in shards:
        s.sort()
    return shards


class RemoteRankFailed(RuntimeError):
    """Another rank hit a fatal error (out of memory, missing library): this rank stops too instead of blocking in the gather."""


def gather_records(local: List[Tuple[int, Optional[List[int]]]], device="cpu",
                   fatal: Optional[BaseException] = None) -> Optional[List[Tuple[int, Optional[List[int]]]]]:
    """local: [(item_idx, token ids or None on error)].  Returns the merged list sorted by item_idx on rank 0
    (None elsewhere).  One all_reduce(MAX) for the record width / count / fatal flag + one all_gather of int32 records.
    `fatal`: the exception that stopped this rank's shard, if any — the flag travels with the size exchange, so EVERY rank raises
    (the failing one its own exception, the others RemoteRankFailed) before anyone enters the all_gather: a multi-GPU eval stops
    instead of hanging until the RCCL watchdog fires."""
    rank, world, _ = world_info()
    if world == 1 or not dist.is_initialized():
        if fatal is not None:
            raise fatal
        return sorted(local, key=lambda r: r[0])
    # the collective's tensors live where the backend works: RCCL ("nccl") on this rank's GPU, gloo on the host — whatever the
    # caller's `device` says (an eval driver started under gloo still passes its cuda device string)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    kmax = max([len(t) for _, t in local if t is not None] + [0])
    meta = torch.tensor([kmax, len(local), 1 if fatal is not None else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    kmax, nmax = int(meta[0]), int(meta[1])
    if int(meta[2]):
        if fatal is not None:
            raise fatal
        raise RemoteRankFailed(f"[rank {rank}] another rank stopped on a fatal error; no records were gathered")
    rec = torch.full((nmax, 2 + kmax), -2, dtype=torch.int32, device=dev)   # -2 = padding row
    for j, (idx, toks) in enumerate(local):
        rec[j, 0] = idx
        if toks is None:
            rec[j, 1] = ERR
        else:
            rec[j, 1] = len(toks)
            if toks:
                rec[j, 2:2 + len(toks)] = torch.tensor(toks, dtype=torch.int32)
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    if rank != 0:
        return None
    merged = []
    for t in out:
        for row in t.cpu().tolist():
            if row[0] == -2 and row[1] == -2:
                continue
            merged.append((row[0], None if row[1] == ERR else row[2:2 + row[1]]))
    merged.sort(key=lambda r: r[0])
    return merged


def request_workers(model, make_generate: Callable, n: Optional[int] = None) -> list:
    """[make_generate(model_i, stream_i)] for `n` requests in flight on one GPU (default: $FO1_INFLIGHT or 2): worker 0 uses the
    loaded model on the current stream, the others a `model.replica()` (shared weights, private KV cache / graphs / scratch)
    on a fresh HIP stream.  `make_generate(model, stream)` must return `generate(i) -> token ids` that runs its device work
    under `torch.cuda.stream(stream)`.  Falls back to one worker when the model cannot be replicated or there is no GPU."""
    if n is None:
        n = int(os.environ.get("FO1_INFLIGHT", "2"))
    if n <= 1 or not torch.cuda.is_available() or not hasattr(model, "replica"):
        return [make_generate(model, torch.cuda.current_stream() if torch.cuda.is_available() else None)]
    workers = [make_generate(model, torch.cuda.current_stream())]
    for _ in range(n - 1):
        workers.append(make_generate(model.replica(), torch.cuda.Stream()))
    return workers


def _fatal(e: BaseException) -> bool:
    """Errors that must stop the evaluation instead of becoming a per-item error record: out-of-memory (every later item would
    fail the same way and the run would silently report wrong metrics — ADVICE r1) and a missing HIP library."""
    if isinstance(e, (MemoryError, torch.cuda.OutOfMemoryError)):
        return True
    msg = str(e).lower()
    return "out of memory" in msg or "hiperroroutofmemory" in msg


def run_sharded(n_items: int, costs: Sequence[float], generate, device="cpu", progress: Optional[Callable] = None, batch: int = 1):
    """Every rank runs `generate` on its shard; rank 0 gets [(i, ids|None)] for all items.

    batch == 1: `generate(i)` -> new token ids of item i.  batch > 1: `generate([i0, i1, ...])` -> [ids per item]: up to `batch`
    items of similar cost go through the engine together (one packed prefill pass, batched decode); when a batched call raises,
    its items are retried one by one so a single bad item costs only its own record.

    `generate` may be a LIST of callables: one worker thread per callable, each pulling the next item / group of the rank's shard
    from a shared queue.  Give every callable its own engine replica + HIP stream (FO1ForCausalLM.replica()).  The records are
    sorted by item index afterwards, so the output ORDER does not depend on the interleaving or the grouping.  The token ids of an
    item are the same in any grouping only as far as the prefill kernels are invariant to the pass's row count: the GEMM tile /
    split-K choice follows M, so a batch-8 pass can differ from a batch-1 pass at bf16 rounding level (a packed pass is bit-identical
    to the one-image passes when the tile is pinned, tests/test_batched_prefill_gpu.py); decode is grouping-invariant.

    A fatal error (out of memory, missing library) on one rank stops ALL ranks: the failing rank raises its exception, the others
    RemoteRankFailed — nobody is left waiting in the gather."""
    rank, world, _ = world_info()
    mine = assign(costs, world)[rank]
    workers = list(generate) if isinstance(generate, (list, tuple)) else [generate]
    if batch > 1:
        order = sorted(mine, key=lambda i: (costs[i], i))       # similar prompt lengths share a pass
        groups = [order[k:k + batch] for k in range(0, len(order), batch)]
    else:
        groups = [[i] for i in mine]
    it = progress(groups) if progress else groups

    def run_group(fn, grp):
        try:
            if batch > 1:
                outs = fn(list(grp))
                return [(i, [int(t) for t in o]) for i, o in zip(grp, outs)]
            return [(grp[0], [int(t) for t in fn(grp[0])])]
        except Exception as e:  # per-item error record instead of the reference's silent `continue` (eval_coco.py:60-65)
            if _fatal(e):
                raise
            if batch > 1 and len(grp) > 1:
                recs = []
                for i in grp:
                    try:
                        recs.append((i, [int(t) for t in fn([i])[0]]))
                    except Exception as e1:
                        if _fatal(e1):
                            raise
                        print(f"[rank {rank}] item {i} failed: {type(e1).__name__}: {e1}")
                        recs.append((i, None))
                return recs
            print(f"[rank {rank}] item {grp[0]} failed: {type(e).__name__}: {e}")
            return [(grp[0], None)]

    fatal: Optional[BaseException] = None
    local: list = []
    if len(workers) == 1:
        try:
            for g in it:
                local.extend(run_group(workers[0], g))
        except Exception as e:      # only fatal ones get here (run_group turns the rest into error records)
            fatal = e
    else:
        import queue
        import threading
        q: "queue.Queue" = queue.Queue()
        for g in it:
            q.put(g)
        lock, errors = threading.Lock(), []

        def loop(fn):
            while True:
                try:
                    g = q.get_nowait()
                except queue.Empty:
                    return
                try:
                    recs = run_group(fn, g)
                except BaseException as e:   # fatal: stop this worker, re-raised on the main thread
                    errors.append(e)
                    return
                with lock:
                    local.extend(recs)

        threads = [threading.Thread(target=loop, args=(fn,), daemon=True) for fn in workers]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            fatal = errors[0]
    local.sort(key=lambda r: r[0])
    return gather_records(local, device, fatal=fatal)
