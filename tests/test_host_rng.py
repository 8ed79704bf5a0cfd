"""The library's restatement of numpy's legacy normal stream (csrc/host_rng.c) against numpy itself:
values and generator state bit for bit.  Host code only -- runs without a GPU."""
import numpy as np
import pytest

from cna_amd.tools import _stats


def _same_state(a, b):
    return a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and np.array_equal(a[1], b[1])


@pytest.mark.parametrize('seed', [0, 1, 12345])
def test_legacy_randn_is_numpys_stream(seed):
    for m, num in ((1, 1), (1, 2), (3, 1), (7, 3), (50, 20), (13, 24), (156, 1), (157, 1), (312, 2), (50, 1000), (3, 16667)):
        for pre in (0, 1, 2, 3, 5, 623, 624, 625):
            np.random.seed(seed)
            if pre:
                np.random.random_sample(pre)
            if pre == 5:
                np.random.randn(1)                       # leaves a cached second value behind
            start = np.random.get_state()
            ref = np.random.randn(m, num)
            ref_state = np.random.get_state()
            ref_next = np.random.randn(3), np.random.randint(0, 1000, 4), np.random.permutation(5)
            np.random.set_state(start)
            got = _stats.legacy_randn(m, num).copy()
            assert got.shape == ref.shape and np.array_equal(got.view(np.uint64), ref.view(np.uint64)), (m, num, pre)
            assert _same_state(np.random.get_state(), ref_state), (m, num, pre)
            nxt = np.random.randn(3), np.random.randint(0, 1000, 4), np.random.permutation(5)
            assert all(np.array_equal(u, v) for u, v in zip(nxt, ref_next))


@pytest.mark.parametrize('threads', [2, 4, 7])
def test_legacy_randn_on_several_threads_is_numpys_stream(threads):
    """The block-parallel stream (csrc/host_rng.c:randn_parallel: every thread derives the key of its own blocks) for
    starting positions that are and are not a multiple of four words from the block end, even and odd counts, a cached
    value pending: values, generator state and what numpy draws next."""
    from cna_amd import _ffi
    lib = _ffi.load()
    lib.cna_host_set_threads(threads)
    _stats._threads_set = True
    try:
        for m, num in ((50, 1000), (1, 16385), (3, 16667), (200, 1000), (7, 9001)):
            for pre in (0, 1, 2, 4, 5, 312, 620, 623, 624, 625, 1248):
                np.random.seed(pre + 17)
                if pre:
                    np.random.random_sample(pre)            # two words each
                if pre == 5:
                    np.random.randn(1)
                start = np.random.get_state()
                ref = np.random.randn(m, num)
                ref_state = np.random.get_state()
                ref_next = np.random.randn(3), np.random.randint(0, 1000, 4)
                np.random.set_state(start)
                got = _stats.legacy_randn(m, num).copy()
                assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), (m, num, pre)
                assert _same_state(np.random.get_state(), ref_state), (m, num, pre)
                nxt = np.random.randn(3), np.random.randint(0, 1000, 4)
                assert all(np.array_equal(u, v) for u, v in zip(nxt, ref_next))
    finally:
        lib.cna_host_set_threads(1)
        _stats._threads_set = False


def test_permutation_draws_match_plain_numpy(monkeypatch):
    """conditional_permutation / grouplevel_permutation through the fast stream == through np.random.randn."""
    rs = np.random.RandomState(3)
    Y = rs.randn(37)
    B = rs.randint(0, 4, 37)
    G = np.repeat(np.arange(13), 3)[:37]
    Yg = rs.randn(13)[G]
    np.random.seed(11)
    a = _stats.conditional_permutation(B, Y, 101)
    ag = _stats.grouplevel_permutation(G, Yg, 55)
    a_next = np.random.rand(2)
    for num in (100, 101):             # freshly seeded: even blocks run on numpy's own state memory
        np.random.seed(11)
        c = _stats.conditional_permutation(B, Y, num, clean=True)
        c_next = np.random.randn(3)
        np.random.seed(11)
        d = _stats.conditional_permutation(B, Y, num)
        assert np.array_equal(c, d) and np.array_equal(c_next, np.random.randn(3))
    np.random.seed(11)
    _stats.conditional_permutation(B, Y, 101)
    assert np.array_equal(_stats.grouplevel_permutation(G, Yg, 55, clean=False), ag)
    monkeypatch.setattr(_stats, 'legacy_randn', lambda m, num, clean=False: np.random.randn(m, num))
    np.random.seed(11)
    b = _stats.conditional_permutation(B, Y, 101)
    bg = _stats.grouplevel_permutation(G, Yg, 55)
    b_next = np.random.rand(2)
    assert np.array_equal(a, b) and np.array_equal(ag, bg) and np.array_equal(a_next, b_next)


def test_other_global_generators_are_left_to_numpy(monkeypatch):
    class Odd:
        _bit_generator = object()
    monkeypatch.setattr(np.random.mtrand, '_rand', Odd(), raising=False)
    out = _stats.legacy_randn(4, 5)
    assert out.shape == (4, 5)


@pytest.mark.parametrize('n,num,nb', [(200, 10000, 1), (200, 2501, 1), (120, 4001, 3), (64, 7000, 2)])
def test_large_draws_take_the_threaded_helpers_and_stay_bit_identical(n, num, nb):
    """Large permutation draws (BASELINE config 5: 200 samples x 10 000 permutations) go through the threaded
    host helpers -- normals' log/sqrt stage split over threads, argsort + gather per column in C --: same
    permuted phenotypes and the same generator state as numpy's randn + argsort, whatever the thread count."""
    from cna_amd import _ffi
    from cna_amd.tools import _stats
    lib = _ffi.load()
    rs = np.random.RandomState(1)
    Y = rs.randn(n)
    B = np.arange(n) % nb
    np.random.seed(7)
    want = np.empty((n, num))
    for b in np.unique(B):
        m = np.flatnonzero(B == b)
        want[m] = Y[m][np.argsort(np.random.randn(len(m), num), axis=0)]
    st_want = np.random.get_state()
    assert n * num >= _stats._BIG_DRAW or nb > 1
    for threads in (1, 3, 8):
        lib.cna_host_set_threads(threads)
        _stats._threads_set = True
        np.random.seed(7)
        got = _stats.conditional_permutation(B, Y, num, clean=True)
        st = np.random.get_state()
        assert np.array_equal(got, want)
        assert st[2] == st_want[2] and np.array_equal(st[1], st_want[1]) and st[3] == st_want[3] and st[4] == st_want[4]
    _stats._threads_set = False


def _reference_conditional_permutation(B, Y, num):
    """reference _stats.py:4-18, verbatim semantics, numpy's own generator"""
    B = np.asarray(B)
    batchind = np.array([np.where(B == b)[0] for b in np.unique(B)], dtype=object)
    ix = np.concatenate([bi[np.argsort(np.random.randn(len(bi), num), axis=0)] for bi in batchind])
    bix = np.zeros((len(Y), num)).astype(int)
    bix[np.concatenate(list(batchind)).astype(int)] = ix
    return Y[bix]


@pytest.mark.parametrize('m,num,levels', [(50, 1000, 1), (50, 1000, 5), (24, 100, 3), (130, 200, 1), (200, 400, 4),
                                          (7, 2, 7), (300, 64, 2)])
def test_native_draw_equals_the_reference_draw(m, num, levels):
    """The whole conditional_permutation on the library's host thread (csrc/host_rng.c:cna_host_draw_start, what
    association() uses for its null): the same permuted phenotypes and the same generator state as the reference's
    np.random.seed + randn + argsort sequence (_association.py:15-16, _stats.py:4-18) -- short columns through the
    rank counts (<= 128 rows per level), long ones through the merge sort."""
    from cna_amd.tools import _stats
    rs = np.random.RandomState(m * 1000 + num + levels)
    Y = rs.randn(m)
    B = rs.randint(0, levels, m).astype(float) if levels > 1 else np.ones(m)
    seed = 17 + m
    h = _stats.native_draw_start(B, Y, num, seed)
    assert h is not None
    table = h.wait()
    after = np.random.get_state()
    nxt = np.random.randn(3)
    np.random.seed(seed)
    want = _reference_conditional_permutation(B, Y, num)
    ref_after = np.random.get_state()
    assert np.array_equal(table[:, 0], Y) and np.array_equal(table[:, 1:], want)
    assert after[2] == ref_after[2] and np.array_equal(after[1], ref_after[1]) and after[3] == ref_after[3] == 0
    assert np.array_equal(nxt, np.random.randn(3))
    # what is not covered falls back (None) without touching the generator
    np.random.seed(5)
    before = np.random.get_state()
    assert _stats.native_draw_start(B, Y, 101, seed) is None and _stats.native_draw_start(B, Y, num, None) is None
    assert np.array_equal(np.random.get_state()[1], before[1])


def test_native_draw_leaves_numpys_generator_alone_until_collected():
    """A draw may be started before the inputs are validated (association() does, for few cells): until wait() numpy's
    global generator is exactly as it was found -- cached second normal included --, abandon() leaves it that way for good,
    and wait() puts it where np.random.seed(seed) + the reference's draws leave it."""
    from cna_amd.tools import _stats
    Y = np.random.RandomState(1).randn(40)
    B = np.ones(40)

    def state():
        st = np.random.get_state()
        return st[1].copy(), st[2], st[3], st[4]

    def same(a, b):
        return np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    np.random.seed(123)
    np.random.randn(3)                                    # an odd count: a second normal is cached (has_gauss = 1)
    found = state()
    assert found[2] == 1
    h = _stats.native_draw_start(B, Y, 200, 9)
    assert h is not None
    assert same(state(), found)                           # started: nothing has happened to the generator
    h.abandon()
    assert same(state(), found)                           # dropped: still nothing
    follow = np.random.randn(5)
    np.random.seed(123)
    np.random.randn(3)
    assert np.array_equal(np.random.randn(5), follow)
    # collected: seed + draws, the cached normal gone as after np.random.seed
    np.random.seed(123)
    np.random.randn(3)
    h = _stats.native_draw_start(B, Y, 200, 9)
    assert same(state(), found)
    table = h.wait()
    got = state()
    np.random.seed(9)
    want = _reference_conditional_permutation(B, Y, 200)
    assert np.array_equal(table[:, 1:], want) and same(state(), got) and got[2] == 0
    # a seed numpy refuses: not covered, the caller's own np.random.seed reports it
    assert _stats.native_draw_start(B, Y, 200, -1) is None and _stats.native_draw_start(B, Y, 200, 'x') is None
    assert same(state(), got)


def test_native_draw_rows_of_no_level_and_ties():
    """NaN batch labels belong to no level: the reference's index matrix stays 0 there (every permutation shows Y[0]);
    exact ties between draws cannot come out of randn, so the rank counts' fall-back is driven directly."""
    from cna_amd.tools import _stats
    Y = np.arange(10, dtype=float)
    B = np.array([0, 0, 1, np.nan, 1, 0, np.nan, 1, 0, 1])
    h = _stats.native_draw_start(B, Y, 50, 3)
    if h is not None:                                     # np.unique keeps NaN as a level of its own: rows with B == NaN are in none
        t = h.wait()
        np.random.seed(3)
        want = np.empty((10, 50))
        want[:] = Y[0]
        for b in np.unique(B):
            mm = np.flatnonzero(B == b)
            if len(mm):
                want[mm] = Y[mm][np.argsort(np.random.randn(len(mm), 50), axis=0)]
            else:
                np.random.randn(0, 50)
        assert np.array_equal(t[:, 1:], want)
    from cna_amd import _ffi
    lib = _ffi.load()
    R = np.array([[1.0, 2.0], [1.0, 0.5], [0.0, 2.0], [1.0, 2.0]])
    y = np.array([10.0, 11.0, 12.0, 13.0])
    out = np.empty((4, 2))
    assert lib.cna_host_argsort_gather(_ffi.ptr(R), 4, 2, _ffi.ptr(y), _ffi.ptr(out), 2, None) == 0
    assert np.array_equal(out, y[np.argsort(R, axis=0, kind='stable')])


def test_native_draw_in_a_forked_child():
    """The draw's worker thread does not survive fork(): a child that inherits 'a worker exists' must start its own
    (pthread_atfork handler in csrc/host_rng.c) instead of waiting for one that is not there."""
    import os
    from cna_amd.tools import _stats
    Y = np.arange(20, dtype=float)
    h = _stats.native_draw_start(np.ones(20), Y, 10, 1)
    assert h is not None
    want = h.wait().copy()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        ok = b'0'
        try:
            import signal
            signal.alarm(20)
            h2 = _stats.native_draw_start(np.ones(20), Y, 10, 1)
            ok = b'1' if h2 is not None and np.array_equal(h2.wait(), want) else b'0'
        finally:
            os.write(w, ok)
            os._exit(0)
    os.close(w)
    got = os.read(r, 1)
    os.waitpid(pid, 0)
    assert got == b'1'


def test_seeded_draw_memo_replays_the_reference_draw():
    """tools/_stats.py:seeded_draw: the permutations of a seeded one-level draw depend on (seed, samples, permutations) only.
    The first draw records which row of y every output is (cna_host_draw_start_idx); later phenotypes with that seed are a
    gather (cna_host_gather_rows).  Values and numpy's generator state equal conditional_permutation's, bit for bit, for
    the draw that fills the memo and for every replay; an abandoned draw leaves the generator alone."""
    from cna_amd.tools import _stats
    from cna_amd.tools._stats import seeded_draw, conditional_permutation
    _stats._draw_memo.clear()
    for m, num, seed in ((50, 1000, 7), (24, 100, 0), (200, 1000, 123), (13, 20, 5)):
        ys = [np.random.RandomState(s).randn(m) for s in (1, 2, 3)]
        kinds = []
        for y in ys:
            y = (y - y.mean()) / y.std()
            np.random.seed(seed)
            want = conditional_permutation(np.ones(m), y, num, clean=True)
            state = np.random.get_state()
            np.random.seed(4242)                       # whatever the generator holds before
            d = seeded_draw(y, num, seed)
            kinds.append(type(d).__name__)
            table = d.wait()
            assert np.array_equal(table[:, 0], y) and np.array_equal(table[:, 1:], want)
            for a, b in zip(np.random.get_state(), state):
                assert np.array_equal(np.asarray(a), np.asarray(b))
        assert kinds == ['NativeDraw', 'ReplayedDraw', 'ReplayedDraw'], kinds
        np.random.seed(99)
        before = np.random.get_state()
        seeded_draw(ys[0], num, seed).abandon()
        for a, b in zip(np.random.get_state(), before):
            assert np.array_equal(np.asarray(a), np.asarray(b))
    assert len(_stats._draw_memo) <= _stats._DRAW_MEMO_ENTRIES
    # not a plain int seed, or an odd permutation count: no memo, the caller's usual draw
    assert seeded_draw(np.zeros(10), 101, 3) is None
