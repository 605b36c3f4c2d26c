"""Continuous batching of the greedy decode loop: ONE decode pool per GPU (llm.DecodePool: 64 / 128
sequence slots advancing through one stream of the weights per step), fed by the packed prefill passes of any number of engine
replicas.  The reference decodes one image at a time (`model.generate`, omchat_qwen2_5_vl.py:143-155 + HF greedy search); here a
prefill pass of up to 32 images hands its sequences to the pool as soon as its first tokens exist and goes on with the next pass while
the pool decodes — sequences of different passes share every decode step, and each leaves when its own stop rule fires.

    svc = PoolService(engine.llm, slots=128)            # a scheduler thread + its HIP stream
    h = svc.submit(replica.llm, seqs, deltas, first_tokens, max_new_tokens, stop_ids)     # after replica.prefill_batch(...)
    h.wait_relocated()                                   # the replica's KV cache may be overwritten by its next pass from here on
    ids = h.result()                                     # [new ids per sequence] (first token included), blocks until all have stopped

Host code only: every tensor op is a libfo1hip.so launch made by DecodePool."""
from __future__ import annotations

import queue
import threading
from typing import List, Optional, Sequence

import torch

from .llm import DecodePool, QwenLLM


class PoolHandle:
    """The sequences of one submit(): `wait_relocated()` orders the caller's stream after the copy of their K / V^T rows out of the
    prefill cache; `result()` -> the generated ids per sequence."""

    def __init__(self, n: int):
        self.n = n
        self._ids: List[Optional[List[int]]] = [None] * n
        self._left = n
        self._done = threading.Event()
        self._joined = threading.Event()
        self._reloc_event: Optional[torch.cuda.Event] = None
        self._error: Optional[BaseException] = None

    def _set(self, k: int, ids: List[int]):
        self._ids[k] = ids
        self._left -= 1
        if self._left == 0:
            self._done.set()

    def _fail(self, e: BaseException):
        self._error = e
        self._joined.set()
        self._done.set()

    def wait_relocated(self, timeout: Optional[float] = None) -> None:
        if not self._joined.wait(timeout):
            raise TimeoutError("decode pool: the sequences were not admitted in time")
        if self._error is not None:
            raise self._error
        torch.cuda.current_stream().wait_event(self._reloc_event)

    def done(self) -> bool:
        return self._done.is_set()

    def result(self, timeout: Optional[float] = None) -> List[List[int]]:
        if not self._done.wait(timeout):
            raise TimeoutError("decode pool: the sequences did not finish in time")
        if self._error is not None:
            raise self._error
        return self._ids


class PoolService:
    """Scheduler thread of one DecodePool.  Loop: admit waiting submissions while slots are free (FIFO, whole submissions) -> enqueue
    `steps_per_round` decode steps (hipGraph replays) -> asynchronous snapshot of the device state -> harvest the PREVIOUS snapshot
    (sequences that had stopped: ids to their handles, slots free again).  The host runs at most two rounds ahead of the GPU and never
    reads a token inside a round."""

    def __init__(self, llm: QwenLLM, slots: int = 128, slot_rows: int = 1024, steps_per_round: int = 4, use_graph: bool = True):
        dev = torch.device(llm.dev)
        self.dev = dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
        self.steps_per_round = max(1, int(steps_per_round))
        self.use_graph = use_graph
        self._args = (llm, slots, slot_rows)
        self.pool: Optional[DecodePool] = None
        self._q: "queue.Queue" = queue.Queue()
        self._stop = False
        self._closed = False
        self._fatal: Optional[BaseException] = None
        self.stats = dict(steps=0, joined=0, finished=0, occupancy_sum=0)
        self._ready = threading.Event()
        self._thread = threading.Thread(target=self._run, name="fo1-decode-pool", daemon=True)
        self._thread.start()
        self._ready.wait()
        if self._fatal is not None:
            raise self._fatal

    # ---- caller side ----------------------------------------------------------------------------------------------------------
    def submit(self, src_llm: QwenLLM, seqs, deltas, first_tokens: torch.Tensor, max_new_tokens: int, stop_ids: Sequence[int] = ()) -> PoolHandle:
        """Hand the sequences of a finished packed prefill (K / V^T in src_llm.kcache / vtcache at the packed rows `seqs`) to the pool.
        The prefill must have been enqueued on the CURRENT stream: an event recorded here orders the pool's relocation after it."""
        if self._fatal is not None:
            raise self._fatal
        if self._closed or self._stop or not self._thread.is_alive():
            raise RuntimeError("decode pool: the service is closed (its scheduler thread has stopped); create a new PoolService")
        n = len(seqs)
        if n > self.pool.P:
            raise ValueError(f"decode pool: {n} sequences in one submission, {self.pool.P} slots")
        h = PoolHandle(n)
        ev = torch.cuda.Event()
        ev.record()
        self._q.put((h, src_llm.kcache, src_llm.vtcache, list(seqs), list(deltas), first_tokens, int(max_new_tokens), tuple(stop_ids), ev))
        if self._closed or not self._thread.is_alive():      # raced close() / the scheduler's fatal path: nobody will ever serve the queue
            self._fail_queued(self._fatal or RuntimeError("decode pool: closed while the submission was being queued"))
        return h

    def close(self):
        """Stops the scheduler after the sequences in flight have finished; whatever is still queued afterwards fails (a
        handle of a closed service must not block forever)."""
        self._stop = True
        self._q.put(None)
        self._thread.join(timeout=60)
        self._closed = True
        self._fail_queued(self._fatal or RuntimeError("decode pool: the service was closed before the submission was admitted"))

    def _fail_queued(self, e: BaseException):
        while True:
            try:
                it = self._q.get_nowait()
            except queue.Empty:
                return
            if it is not None:
                it[0]._fail(e)

    # ---- scheduler thread -----------------------------------------------------------------------------------------------------
    def _admit(self, pool: DecodePool, waiting: list) -> None:
        while waiting:
            h, kc, vt, seqs, deltas, first, max_new, stop_ids, ev = waiting[0]
            if len(seqs) > len(pool.free):
                return
            if not pool.can_take(stop_ids):
                return                                   # every row of the stop-set table is in use by live sequences (32 different sets)
            if pool.live and not pool.fits(seqs, max_new):
                return                                   # longer slots than the pool has: its caches can only grow while it is empty
            waiting.pop(0)
            try:
                torch.cuda.current_stream().wait_event(ev)       # the prefill that produced the rows and the first tokens
                pool.join(kc, vt, seqs, deltas, first, max_new, stop_ids, tags=[(h, k) for k in range(len(seqs))])
                h._reloc_event = torch.cuda.Event()
                h._reloc_event.record()
                h._joined.set()
                self.stats["joined"] += len(seqs)
            except BaseException as e:      # a bad submission fails its own handle, not the pool
                h._fail(e)

    def _run(self):
        try:
            torch.cuda.set_device(self.dev)
            self.stream = torch.cuda.Stream(device=self.dev)
            with torch.cuda.stream(self.stream):
                self.pool = DecodePool(*self._args)
            self.stream.synchronize()
        except BaseException as e:
            self._fatal = e
            self._ready.set()
            return
        self._ready.set()
        pool = self.pool
        waiting: list = []
        prev = None
        pinned = [None, None]
        rnd = 0
        try:
            with torch.cuda.stream(self.stream):
                while True:
                    # new submissions: block only when there is nothing to decode
                    block = not pool.live and not waiting and prev is None
                    while True:
                        try:
                            item = self._q.get(block=block)
                        except queue.Empty:
                            break
                        if item is None:
                            self._stop = True
                            break
                        waiting.append(item)
                        block = False
                    if self._stop and not pool.live and not waiting:
                        return
                    self._admit(pool, waiting)
                    snap = None
                    if pool.live:
                        for _ in range(self.steps_per_round):
                            pool.step(self.use_graph)
                        self.stats["steps"] += self.steps_per_round
                        self.stats["occupancy_sum"] += self.steps_per_round * len(pool.live)
                        snap = pool.snapshot(pinned[rnd & 1])
                        pinned[rnd & 1] = snap[4]
                        rnd += 1
                    if prev is not None:
                        for _, (h, k), ids in pool.harvest(prev):
                            h._set(k, ids)
                            self.stats["finished"] += 1
                    prev = snap
        except BaseException as e:          # a failing step is fatal for every sequence in flight
            self._fatal = e
            for it in waiting:
                it[0]._fail(e)
            if pool is not None:
                for tag in list(pool.live.values()):
                    tag[0]._fail(e)
            while True:
                try:
                    it = self._q.get_nowait()
                except queue.Empty:
                    break
                if it is not None:
                    it[0]._fail(e)


class PoolGroup:
    """Several decode pools on ONE GPU behind the PoolService interface (round 6).  A pool step is a chain of ~290 short dependent launches,
    each with a ramp and a tail and none using more than ~40 % of the memory system; two such chains on two HIP streams interleave on the
    GPU, and the second reader of a weight tile finds it in the 256 MB Infinity Cache while the chains stay within a layer of each other:
    two pools of 128 live sequences stepping concurrently deliver 44.5k tokens/s where one delivers 32.9k (profiles/r06_two_decode_pools.json).
    Every pool is a full PoolService (own scheduler thread, stream, DecodePool); a submission goes, whole, to the pool with the fewest
    sequences in flight.  Per sequence nothing changes: same kernels, same launch shapes, same sums — the ids do not depend on which pool
    or which neighbours a sequence has (tests/test_decode_pool_gpu.py)."""

    def __init__(self, llm: QwenLLM, pools: int = 2, slots: int = 128, slot_rows: int = 1024, steps_per_round: int = 4, use_graph: bool = True):
        if pools < 1:
            raise ValueError("a pool group has at least one pool")
        self.services = [PoolService(llm, slots=slots, slot_rows=slot_rows, steps_per_round=steps_per_round, use_graph=use_graph) for _ in range(pools)]
        self._submitted = [0] * pools         # sequences handed to each pool so far (its own statistics count the finished ones)
        self._lock = threading.Lock()

    @property
    def pool(self) -> DecodePool:
        return self.services[0].pool

    @property
    def pools(self) -> List[DecodePool]:
        return [s.pool for s in self.services]

    @property
    def stats(self) -> dict:
        out = dict(steps=0, joined=0, finished=0, occupancy_sum=0)
        for s in self.services:
            for k in out:
                out[k] += s.stats[k]
        out["pools"] = len(self.services)
        return out

    def submit(self, src_llm: QwenLLM, seqs, deltas, first_tokens: torch.Tensor, max_new_tokens: int, stop_ids: Sequence[int] = ()) -> PoolHandle:
        with self._lock:
            k = min(range(len(self.services)), key=lambda i: (self._submitted[i] - self.services[i].stats["finished"], i))
            self._submitted[k] += len(seqs)
        return self.services[k].submit(src_llm, seqs, deltas, first_tokens, max_new_tokens, stop_ids)

    def close(self):
        for s in self.services:
            s.close()
