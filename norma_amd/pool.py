"""Refill policy of the decode pool (include/norma_hip.h: nh_pool_*).

The reference's decode loop ends per sequence at eot (src/models/whisper/model.rs:317) and it decodes one stream at a time
(src/lib.rs:462-464); a batch decoded in lockstep makes the short sequences wait for the longest one.  The pool keeps a fixed
number of decode rows busy instead: the encoder is still fed `staging` clips at a time, every encoded clip is admitted to
whichever row is free, finished rows are collected every `check_every` steps.  The policy is engine-agnostic (it only calls
the five methods below), so the CPU tests drive it with a counting stand-in and the GPU tests with HipWhisper.

engine methods used: pool_begin(rows, max_new, per_clip_language), pool_admit(src_row, dst_row, lang),
pool_step(n) -> flags per row (0 running, 1 / 2 finished, 3 empty), pool_collect(rows) -> [result],
and the caller's encode(first_clip, n_clips, row0, must) which must leave clips first .. first + n - 1 encoded in rows
row0 ... -- or, when `must` is false, may return False to say "the encoder is busy, ask again" (several pools share one GPU:
the pool then goes on decoding what it has instead of waiting for the encoder with its rows idle).
"""
from typing import Callable, List, Optional, Sequence


class DecodePool:
    def __init__(self, engine, rows: int = 64, staging: int = 32, max_new_tokens: int = 0, check_every: int = 16,
                 per_clip_language: bool = False):
        assert rows >= 1 and staging >= 1 and check_every >= 1
        self.e, self.rows, self.staging, self.check_every = engine, rows, staging, check_every
        self.max_new, self.per_clip_language = max_new_tokens, per_clip_language
        self.steps = 0          # decode steps launched
        self.row_steps = 0      # sum over steps of the rows that were busy (what the step kernels' per-row work scales with)
        self.encodes = 0

    def run(self, n_clips: int, encode: Callable[[int, int, int, bool], Optional[bool]], langs: Optional[Sequence[int]] = None,
            on_result: Optional[Callable[[int, dict], None]] = None) -> List[dict]:
        """Decode clips 0 .. n_clips - 1; returns their results in clip order."""
        e, R = self.e, self.rows
        e.pool_begin(R, self.max_new, self.per_clip_language)
        results: List[Optional[dict]] = [None] * n_clips
        owner = [-1] * R                  # clip decoding in each row
        staged: List[int] = []            # clips encoded and waiting for a row, in order; clip c sits in staging row R + (c - staged_first)
        staged_first = 0
        next_clip = 0
        busy = 0
        while next_clip < n_clips or staged or busy:
            if not staged and next_clip < n_clips:
                n = min(self.staging, n_clips - next_clip)
                if encode(next_clip, n, R, busy == 0) is not False:   # busy == 0: nothing to decode meanwhile, wait for the encoder
                    self.encodes += 1
                    staged_first = next_clip
                    staged = list(range(next_clip, next_clip + n))
                    next_clip += n
            for r in range(R):            # admit in clip order into the lowest free rows
                if not staged:
                    break
                if owner[r] < 0:
                    c = staged.pop(0)
                    e.pool_admit(R + (c - staged_first), r, -1 if langs is None else int(langs[c]))
                    owner[r] = c
                    busy += 1
            flags = e.pool_step(self.check_every)
            self.steps += self.check_every
            self.row_steps += busy * self.check_every
            fin = [r for r in range(R) if owner[r] >= 0 and flags[r] in (1, 2)]
            if fin:
                for r, res in zip(fin, e.pool_collect(fin)):
                    results[owner[r]] = res
                    if on_result:
                        on_result(owner[r], res)
                    owner[r] = -1
                    busy -= 1
        return results  # type: ignore[return-value]


class FedDecodePool:
    """One decoding context fed by encoder contexts of the same weight set (nh_pool_admit_from): the pool never stalls for an
    encoder submission -- its stream only ever runs decode steps and the device-to-device moves of admitted clips -- while an
    encoder thread keeps the encoder contexts busy one after the other.  `encode(e, first_clip, n)` must leave clips first ..
    first + n - 1 encoded in rows 0 .. n - 1 of encoder context e (called on the encoder thread; an encoder context is handed
    out again only when every clip of its previous submission has been admitted)."""

    def __init__(self, engine, encoders: Sequence, rows: int = 64, batch: int = 32, max_new_tokens: int = 0, check_every: int = 16,
                 per_clip_language: bool = False):
        assert rows >= 1 and batch >= 1 and check_every >= 1 and len(encoders) >= 1
        self.e, self.encoders, self.rows, self.batch, self.check_every = engine, list(encoders), rows, batch, check_every
        self.max_new, self.per_clip_language = max_new_tokens, per_clip_language
        self.steps = self.row_steps = self.encodes = 0

    def run(self, n_clips: int, encode: Callable[[int, int, int], None], langs: Optional[Sequence[int]] = None) -> List[dict]:
        import queue
        import threading
        e, R, NE = self.e, self.rows, len(self.encoders)
        e.pool_begin(R, self.max_new, self.per_clip_language)
        ready: "queue.Queue" = queue.Queue()                 # (encoder index, first clip, n) in submission order
        free = [threading.Semaphore(1) for _ in range(NE)]   # encoder context not holding un-admitted clips
        errs: List[BaseException] = []

        def feeder():
            try:
                for k, first in enumerate(range(0, n_clips, self.batch)):
                    i = k % NE
                    free[i].acquire()
                    n = min(self.batch, n_clips - first)
                    encode(i, first, n)
                    self.encodes += 1
                    ready.put((i, first, n))
            except BaseException as ex:   # noqa: BLE001 -- re-raised on the calling thread
                errs.append(ex)
            finally:
                ready.put(None)
        th = threading.Thread(target=feeder)
        th.start()
        results: List[Optional[dict]] = [None] * n_clips
        owner = [-1] * R
        cur = None          # submission being admitted: [encoder index, first clip, n, next index in it]
        fed_all, busy, done_clips = False, 0, 0
        try:
            while done_clips < n_clips and not errs:
                while True:                      # admit what is encoded into the lowest free rows, in clip order
                    if cur is None and not fed_all:
                        try:
                            item = ready.get(block=(busy == 0))    # nothing to decode: wait for the encoder
                        except queue.Empty:
                            break
                        if item is None:
                            fed_all = True
                            break
                        cur = [item[0], item[1], item[2], 0]
                    if cur is None:
                        break
                    r = next((r for r in range(R) if owner[r] < 0), None)
                    if r is None:
                        break
                    i, first, n, j = cur
                    c = first + j
                    e.pool_admit_from(self.encoders[i], j, r, -1 if langs is None else int(langs[c]))
                    owner[r] = c
                    busy += 1
                    cur[3] += 1
                    if cur[3] == n:
                        free[i].release()        # every clip of that submission has been moved: the context may encode again
                        cur = None
                if busy == 0:
                    continue
                flags = e.pool_step(self.check_every)
                self.steps += self.check_every
                self.row_steps += busy * self.check_every
                fin = [r for r in range(R) if owner[r] >= 0 and flags[r] in (1, 2)]
                if fin:
                    for r, res in zip(fin, e.pool_collect(fin)):
                        results[owner[r]] = res
                        owner[r] = -1
                        busy -= 1
                        done_clips += 1
        finally:
            for s_ in free:                      # let a feeder that is waiting for a context run to its end
                s_.release()
            th.join()
        if errs:
            raise errs[0]
        return results  # type: ignore[return-value]
