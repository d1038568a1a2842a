"""CPU: the decode pool's refill policy (norma_amd/pool.py) against a counting stand-in for the engine -- every clip is
admitted once into a free row, collected once after it finished, results come back in clip order, the encoder is fed
`staging` clips at a time.  (The reference decodes one stream at a time, src/lib.rs:462-464, and ends each sequence at eot,
model.rs:317: no counterpart; the GPU side is tests/test_gpu_pool.py.)"""
import numpy as np
import pytest

from norma_amd import pool


class FakeEngine:
    def __init__(self, lengths, prompt=3):
        self.lengths, self.prompt = lengths, prompt
        self.log = []
        self.busy_every = 0

    def pool_begin(self, rows, max_new, per_clip_language):
        self.rows = rows
        self.left = [None] * rows          # steps still to run per row (None: empty)
        self.clip = [None] * rows
        self.staged = {}
        self.lang = [None] * rows
        self.max_concurrent = 0

    def encode(self, first, n, row0, must):
        assert row0 == self.rows
        self.asked = getattr(self, "asked", 0) + 1
        if self.busy_every and not must and self.asked % self.busy_every:
            return False                   # "the encoder is busy": the pool must go on decoding and ask again
        self.staged = {row0 + i: first + i for i in range(n)}
        self.log.append(("encode", first, n))

    def pool_admit(self, src, dst, lang):
        assert self.left[dst] is None and src in self.staged
        c = self.staged.pop(src)
        self.clip[dst], self.left[dst], self.lang[dst] = c, self.lengths[c] + self.prompt - 1, lang
        self.log.append(("admit", c, dst))

    def pool_step(self, n):
        self.max_concurrent = max(self.max_concurrent, sum(l is not None for l in self.left))
        flags = np.zeros(self.rows, dtype=np.int32)
        for r in range(self.rows):
            if self.left[r] is None:
                flags[r] = 3
            else:
                self.left[r] = max(0, self.left[r] - n)
                flags[r] = 1 if self.left[r] == 0 else 0
        return flags

    def pool_collect(self, rows):
        out = []
        for r in rows:
            assert self.left[r] == 0
            out.append(dict(clip=self.clip[r], lang=self.lang[r]))
            self.left[r] = None
        return out


@pytest.mark.parametrize("rows,staging,check", [(4, 3, 2), (8, 8, 16), (2, 5, 1), (64, 32, 16)])
def test_every_clip_is_admitted_once_collected_once_and_returned_in_clip_order(rows, staging, check):
    rng = np.random.default_rng(rows * 100 + staging)
    N = 57
    lengths = [int(x) for x in rng.choice([3, 10, 40, 41, 120], size=N)]
    e = FakeEngine(lengths)
    e.busy_every = 3 if rows % 2 == 0 else 0   # half of the cases: an encoder that says "busy" two times out of three
    dp = pool.DecodePool(e, rows=rows, staging=staging, check_every=check)
    seen = []
    res = dp.run(N, e.encode, langs=list(range(1000, 1000 + N)), on_result=lambda c, r: seen.append(c))
    assert [r["clip"] for r in res] == list(range(N)) and [r["lang"] for r in res] == list(range(1000, 1000 + N))
    assert sorted(seen) == list(range(N))
    assert [x[1] for x in e.log if x[0] == "admit"] == list(range(N))          # admitted in clip order
    enc = [x for x in e.log if x[0] == "encode"]
    assert [x[1] for x in enc] == list(range(0, N, staging)) and sum(x[2] for x in enc) == N
    assert dp.encodes == len(enc) and e.max_concurrent <= rows
    assert dp.row_steps >= sum(lengths) and dp.steps % check == 0
    if N > rows + staging:
        assert e.max_concurrent == rows                                        # the pool does fill up


def test_pool_keeps_rows_busy_where_lockstep_batches_wait_for_the_longest():
    lengths = [300 if k % 16 == 0 else 20 for k in range(256)]
    e = FakeEngine(lengths, prompt=1)
    dp = pool.DecodePool(e, rows=32, staging=32, check_every=4)
    dp.run(len(lengths), e.encode)
    lockstep = sum(32 * max(lengths[g:g + 32]) for g in range(0, len(lengths), 32))
    assert dp.row_steps < 0.5 * lockstep


class FakeEncoder:
    def __init__(self):
        self.rows = {}


class FedEngine(FakeEngine):
    def pool_admit_from(self, enc, src, dst, lang):
        assert self.left[dst] is None and src in enc.rows
        c = enc.rows.pop(src)
        self.clip[dst], self.left[dst], self.lang[dst] = c, self.lengths[c] + self.prompt - 1, lang
        self.log.append(("admit", c, dst))


@pytest.mark.parametrize("rows,batch,n_enc,check", [(8, 5, 2, 4), (64, 32, 2, 16), (3, 7, 1, 1), (16, 4, 3, 2)])
def test_one_pool_fed_by_encoder_contexts_admits_every_clip_once_in_order_and_reuses_a_context_only_when_it_is_drained(rows, batch, n_enc, check):
    rng = np.random.default_rng(rows + batch)
    N = 83
    lengths = [int(x) for x in rng.choice([3, 10, 40, 120], size=N)]
    e, encs = FedEngine(lengths), [FakeEncoder() for _ in range(n_enc)]
    submissions = []

    def encode(i, first, n):
        assert not encs[i].rows                      # handed out again only when every clip of its last submission was admitted
        encs[i].rows = {k: first + k for k in range(n)}
        submissions.append((i, first, n))
    fp = pool.FedDecodePool(e, encs, rows=rows, batch=batch, check_every=check)
    res = fp.run(N, encode, langs=list(range(500, 500 + N)))
    assert [r["clip"] for r in res] == list(range(N)) and [r["lang"] for r in res] == list(range(500, 500 + N))
    assert [x[1] for x in e.log if x[0] == "admit"] == list(range(N))
    assert [sbm[1] for sbm in submissions] == list(range(0, N, batch)) and [sbm[0] for sbm in submissions] == [k % n_enc for k in range(len(submissions))]
    assert fp.encodes == len(submissions) and fp.row_steps >= sum(lengths) and e.max_concurrent <= rows


def test_an_error_on_the_encoder_thread_reaches_the_caller():
    e, encs = FedEngine([5] * 20), [FakeEncoder()]

    def encode(i, first, n):
        if first >= 8:
            raise RuntimeError("encoder failed")
        encs[i].rows = {k: first + k for k in range(n)}
    with pytest.raises(RuntimeError):
        pool.FedDecodePool(e, encs, rows=4, batch=4, check_every=2).run(20, encode)
