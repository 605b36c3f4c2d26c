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

# What the last run_sharded() of this process did (bench.py's `scale` block reads it): items and seconds of this rank's shard, the
# gather's milliseconds, bytes and backend, the world size — and, on rank 0, the merged records themselves.
LAST: dict = {}


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
    import time
    t_meet = time.perf_counter()
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    kmax, nmax = int(meta[0]), int(meta[1])          # (the host read waits for the collective: this rank has now MET the slowest one)
    LAST.update(gather_wait_ms=(time.perf_counter() - t_meet) * 1e3)
    t_coll = time.perf_counter()
    if int(meta[2]):
        if fatal is not None:
            raise fatal
        raise RemoteRankFailed(f"[rank {rank}] another rank stopped on a fatal error; no records were gathered")
    # the record block is built on the HOST and crosses to the device in one copy (row-by-row device writes were three tiny H2D copies
    # per record, ~2 000 per rank on COCO-5k: VERDICT r4 weak #14)
    import numpy as np
    host = np.full((nmax, 2 + kmax), -2, dtype=np.int32)                     # -2 = padding row
    for j, (idx, toks) in enumerate(local):
        host[j, 0] = idx
        if toks is None:
            host[j, 1] = ERR
        else:
            host[j, 1] = len(toks)
            if toks:
                host[j, 2:2 + len(toks)] = np.asarray(toks, dtype=np.int32)
    rec = torch.from_numpy(host).to(dev)
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    if dev.type == "cuda":
        torch.cuda.current_stream().synchronize()
    LAST.update(gather_record_bytes_per_rank=int(host.nbytes), gather_backend=dist.get_backend(), gather_world=dist.get_world_size(),
                gather_collective_ms=(time.perf_counter() - t_coll) * 1e3)      # record block H2D + the one all_gather, after the ranks have met
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
    under `torch.cuda.stream(stream)`.  Falls back to one worker when the model cannot be replicated or there is no GPU.

    $FO1_DECODE_POOL (default 128; 0 = off): every worker hands the sequences of its prefill passes to ONE decode pool of that many
    slots (vlm_fo1_amd/serving.py: continuous batching) instead of decoding its own group — measured on the COCO-shaped loop
    (bench.py `driver_level`).  Replicas and their streams are kept on the model, so a second evaluation in the same process re-uses
    the captured graphs."""
    if n is None:
        n = int(os.environ.get("FO1_INFLIGHT", "2"))
    if not torch.cuda.is_available() or not hasattr(model, "replica"):
        return [make_generate(model, torch.cuda.current_stream() if torch.cuda.is_available() else None)]
    pool = int(os.environ.get("FO1_DECODE_POOL", "128"))
    eng = getattr(model, "engine", None)
    if eng is not None and hasattr(eng, "enable_decode_pool"):
        if pool in (64, 128):
            eng.enable_decode_pool(slots=pool)
        elif getattr(eng, "_pool_svc", None) is not None:
            eng.disable_decode_pool()
    if n <= 1:
        return [make_generate(model, torch.cuda.current_stream())]
    kept = model.__dict__.setdefault("_worker_replicas", [])
    while len(kept) < n - 1:
        kept.append((model.replica(), torch.cuda.Stream()))
    for r, _ in kept:                       # replicas made before the pool was switched follow the model
        if getattr(r, "engine", None) is not None and eng is not None:
            r.engine._pool_svc = getattr(eng, "_pool_svc", None)
    return [make_generate(model, torch.cuda.current_stream())] + [make_generate(r, st) for r, st in kept[:n - 1]]


def item_cost(width: int, height: int, n_boxes: int, aux: str = "dynamic", new_tokens: int = 64) -> float:
    """Estimated work of one item in GFLOP-equivalents from (pixels, N) — SURVEY §8e's cost key.  Mirrors the path's own sizing:
    long side capped at 2048 (mm_utils.py:447-455), smart-resize to multiples of 28 (S patches), `dynamic` aux = the resized image
    or `squash` 768x768; the coefficients are the dense flop of each stage per unit: ViT 1.26 GF per patch (+ full
    attention), DaViT-L 2.5 MF per aux pixel, SimpleFPN 0.13 GF per patch, LLM prefill 5.5 GF per row (S/4 image rows + 2 per
    box + ~60 text), decode `new_tokens` x 6.2 GB of weights priced at the 400 flop/byte machine balance."""
    from vlm_fo1.model.image_processing import smart_resize
    w, h = max(int(width), 1), max(int(height), 1)
    if max(w, h) > 2048:
        r = 2048.0 / max(w, h)
        w, h = max(int(w * r), 1), max(int(h * r), 1)
    try:
        rh, rw = smart_resize(h, w, 28, 56 * 56, 2048 * 2048)
    except ValueError:
        rh, rw = max(28, h // 28 * 28), max(28, w // 28 * 28)
    S = (rh // 14) * (rw // 14)
    aux_px = 768 * 768 if aux == "squash" else w * h
    rows = S // 4 + 2 * int(n_boxes) + 60
    vit = 1.26 * S + 4 * (4.0 * S * S * 1280) / 1e9          # 28 windowed blocks are linear in S; 4 full-attention blocks are not
    llm = 5.5 * rows + 36 * (2.0 * rows * rows * 2048) / 1e9
    return vit + 2.5e-3 * aux_px + 0.13 * S + llm + new_tokens * 6.2 * 0.4


def image_size(path: str, boxes=None):
    """(width, height) from the image header (PIL reads it lazily, no decode); when the file cannot be opened: the extent of the
    item's boxes, else a COCO-typical 640 x 480 — the cost model must never stop an evaluation."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            return im.size
    except Exception:
        if boxes:
            return max(int(max(b[2] for b in boxes)), 1), max(int(max(b[3] for b in boxes)), 1)
        return 640, 480


class Deferred:
    """What a batched `generate(idxs, prepared)` may return instead of the ids: the group's prefill has run and its sequences are
    decoding in the shared pool; `result()` blocks until they have stopped and returns [ids per item].  run_sharded keeps a few of them
    outstanding per worker, so two workers keep the GPU as busy as the closed-loop bench (bench.py `end_to_end`)."""

    def __init__(self, resolve: Callable):
        self._resolve = resolve

    def result(self):
        return self._resolve()


class Prefetcher:
    """Runs `prepare(i)` (PIL decode / resize, tokenisation, uploads: host work of a1) on helper threads up to `depth` items ahead of
    the consumer, in the order the items will be consumed — the per-rank prefetch thread of SURVEY §8e: at > 100 images/s per GPU the
    host side of an item costs more than its share of a packed pass.  `get(i)` returns prepare(i)'s result or re-raises its
    exception (so a bad image becomes that item's error record, as before).  Results are handed over once; memory is bounded by
    `depth` prepared items."""

    def __init__(self, prepare: Callable, order: Sequence[int], depth: int = 16, threads: int = 2):
        from concurrent.futures import ThreadPoolExecutor
        import threading
        self._prepare, self._order, self._depth = prepare, list(order), max(1, int(depth))
        # helper threads start on device 0 whatever the creating thread uses: on rank r > 0 the device-side preprocessing would launch
        # on a stream of the wrong GPU — every helper first selects the creator's device
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(threads)), thread_name_prefix="fo1-prefetch",
                                        initializer=(lambda: torch.cuda.set_device(dev)) if dev is not None else None)
        self._fut, self._next, self._taken = {}, 0, 0
        self._lock = threading.Lock()
        self._fill()

    def _fill(self):
        with self._lock:
            while self._next < len(self._order) and self._next - self._taken < self._depth:
                i = self._order[self._next]
                if i not in self._fut:
                    self._fut[i] = self._pool.submit(self._prepare, i)
                self._next += 1

    def get(self, i: int):
        with self._lock:
            f = self._fut.pop(i, None)
            if f is not None:              # only a scheduled item frees a slot of the window: a retry (its future was consumed by
                self._taken += 1           # the failed group) must not widen it
        self._fill()
        if f is None:                      # not scheduled (a retry after a failed group, or out-of-order use): do it now
            return self._prepare(i)
        return f.result()

    def close(self):
        self._pool.shutdown(wait=False, cancel_futures=True)


def _fatal(e: BaseException) -> bool:
    """Errors that must stop the evaluation instead of becoming a per-item error record: out-of-memory (every later item would
    fail the same way and the run would silently report wrong metrics — ADVICE r1) and a missing HIP library."""
    if isinstance(e, (MemoryError, torch.cuda.OutOfMemoryError)):
        return True
    msg = str(e).lower()
    return "out of memory" in msg or "hiperroroutofmemory" in msg


def run_sharded(n_items: int, costs: Sequence[float], generate, device="cpu", progress: Optional[Callable] = None, batch: int = 1,
                prepare: Optional[Callable] = None, prefetch_depth: int = 16, prefetch_threads: int = 2):
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
    RemoteRankFailed — nobody is left waiting in the gather.

    `prepare(i)`: optional host-side preparation of item i (image decode / resize / tokenisation / upload).  When given it runs on
    a Prefetcher ahead of the GPU work and `generate` receives the prepared object(s) as a second argument: `generate(i, prepared)`
    or `generate([i0, ...], [prepared0, ...])`."""
    import time
    t_start = time.perf_counter()
    enter_wall = time.time()
    rank, world, _ = world_info()
    mine = assign(costs, world)[rank]
    workers = list(generate) if isinstance(generate, (list, tuple)) else [generate]
    if batch > 1:
        order = sorted(mine, key=lambda i: (costs[i], i))       # similar prompt lengths share a pass
        groups = [order[k:k + batch] for k in range(0, len(order), batch)]
    else:
        groups = [[i] for i in mine]
    it = progress(groups) if progress else groups
    pf = Prefetcher(prepare, [i for g in groups for i in g], prefetch_depth, prefetch_threads) if prepare is not None else None

    def call(fn, grp, single):
        if pf is None:
            return fn(grp[0]) if single else fn(list(grp))
        return fn(grp[0], pf.get(grp[0])) if single else fn(list(grp), [pf.get(i) for i in grp])

    def retry_one_by_one(fn, grp):
        recs = []
        for i in grp:
            try:
                o = call(fn, [i], False)
                o = o.result() if isinstance(o, Deferred) else o
                recs.append((i, [int(t) for t in o[0]]))
            except Exception as e1:
                if _fatal(e1):
                    raise
                print(f"[rank {rank}] item {i} failed: {type(e1).__name__}: {e1}")
                recs.append((i, None))
        return recs

    def failed(fn, grp, e):
        """per-item error record instead of the reference's silent `continue` (eval_coco.py:60-65)"""
        if _fatal(e):
            raise e
        if batch > 1 and len(grp) > 1:
            return retry_one_by_one(fn, grp)
        print(f"[rank {rank}] item {grp[0]} failed: {type(e).__name__}: {e}")
        return [(grp[0], None)]

    def start_group(fn, grp):
        """-> records, or (grp, Deferred) when the worker handed the group's sequences to the decode pool and can go on with its next
        group while they decode (`generate` returned a Deferred)."""
        try:
            if batch > 1:
                outs = call(fn, grp, False)
                if isinstance(outs, Deferred):
                    return grp, outs
                return [(i, [int(t) for t in o]) for i, o in zip(grp, outs)]
            return [(grp[0], [int(t) for t in call(fn, grp, True)])]
        except Exception as e:
            return failed(fn, grp, e)

    def settle(fn, pending):
        grp, d = pending
        try:
            return [(i, [int(t) for t in o]) for i, o in zip(grp, d.result())]
        except Exception as e:
            return failed(fn, grp, e)

    def run_worker(fn, next_group, emit):
        """One worker's loop: up to OUTSTANDING deferred groups in flight (their prefill done, their sequences decoding in the pool)."""
        import collections
        pending = collections.deque()
        while True:
            g = next_group()
            if g is None:
                break
            r = start_group(fn, g)
            if isinstance(r, tuple):
                pending.append(r)
                if len(pending) > OUTSTANDING:
                    emit(settle(fn, pending.popleft()))
            else:
                emit(r)
        while pending:
            emit(settle(fn, pending.popleft()))

    OUTSTANDING = 2
    fatal: Optional[BaseException] = None
    local: list = []
    if len(workers) == 1:
        try:
            gi = iter(it)
            run_worker(workers[0], lambda: next(gi, None), local.extend)
        except Exception as e:      # only fatal ones get here (the rest became error records)
            fatal = e
    else:
        import queue
        import threading
        q: "queue.Queue" = queue.Queue()
        for g in it:
            q.put(g)
        lock, errors = threading.Lock(), []

        def next_group():
            try:
                return q.get_nowait()
            except queue.Empty:
                return None

        def emit(recs):
            with lock:
                local.extend(recs)

        def loop(fn):
            try:
                run_worker(fn, next_group, emit)
            except BaseException as e:   # fatal: stop this worker, re-raised on the main thread
                errors.append(e)

        threads = [threading.Thread(target=loop, args=(fn,), daemon=True) for fn in workers]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            fatal = errors[0]
    if pf is not None:
        pf.close()
    local.sort(key=lambda r: r[0])
    import time
    t_shard = time.perf_counter()
    LAST.clear()
    LAST.update(rank=rank, world=world, shard_items=len(mine), shard_seconds=t_shard - t_start, enter_wall=enter_wall,
                shard_cost=float(sum(costs[i] for i in mine)))
    merged = gather_records(local, device, fatal=fatal)
    LAST.update(gather_ms=(time.perf_counter() - t_shard) * 1e3, merged=merged)
    return merged
