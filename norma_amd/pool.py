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
