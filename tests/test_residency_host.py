"""Host logic of the windowed residency (models/residency.py): page ownership, window layout, the step stream.  Pure index
arithmetic -- every rank computes every rank's stream without communication, so it has to be deterministic and exact."""
import numpy as np
import pytest

from gnn_tableextraction_amd.models import residency as R


def test_page_owner_deals_pages_evenly_and_reproducibly():
    for n, w in ((1000, 8), (7, 2), (5, 8), (64, 1)):
        o = R.page_owner(n, w, seed=3)
        assert o.shape == (n,) and o.min() >= 0 and o.max() < w
        counts = np.bincount(o, minlength=w)
        assert counts.max() - counts.min() <= 1                     # round-robin over a shuffle
        np.testing.assert_array_equal(o, R.page_owner(n, w, seed=3))
    assert not np.array_equal(R.page_owner(1000, 8, seed=3), R.page_owner(1000, 8, seed=4))


def test_window_ranges_partition_the_pages_within_the_slot():
    rng = np.random.default_rng(0)
    nodes = rng.integers(20, 2000, 500)
    per_node = 5000.0
    slot = 40 * 2000 * per_node
    rs = R.window_ranges(nodes, per_node, slot)
    assert rs[0][0] == 0 and rs[-1][1] == len(nodes)
    for (a0, a1), (b0, b1) in zip(rs, rs[1:]):
        assert a1 == b0 and a0 < a1
    cap = int(slot // per_node)
    for p0, p1 in rs:
        assert nodes[p0:p1].sum() <= cap
    # as few windows as the slot allows, and balanced: no window -- the last one in particular -- is a small remainder
    assert len(rs) == -(-int(nodes.sum()) // cap) or len(rs) == -(-int(nodes.sum()) // cap) + 1
    per_win = np.array([nodes[p0:p1].sum() for p0, p1 in rs])
    assert per_win.min() >= 0.8 * per_win.max()
    bounded = R.window_ranges(nodes, per_node, slot, max_nodes=10_000)   # the row-map bound is the tighter one
    assert all(nodes[p0:p1].sum() <= 10_000 for p0, p1 in bounded) and len(bounded) > len(rs)
    with pytest.raises(ValueError):
        R.window_ranges(np.array([50, 5000, 50]), per_node, 1000 * per_node)


def test_layout_is_the_same_arithmetic_for_every_caller():
    rng = np.random.default_rng(1)
    nodes, edges = rng.integers(20, 2000, 300), rng.integers(100, 12000, 300)
    a = R.WindowedPages.layout(nodes, edges, 831, 2e9, True)
    b = R.WindowedPages.layout(nodes.copy(), edges.copy(), 831, 2e9, True)
    assert a == b and a[0][0] == 0 and a[-1][1] == 300
    per = R.WindowedPages.bytes_per_node(nodes, edges, 831, True)
    assert per > 96 * 52                                              # the image row alone
    assert R.WindowedPages.bytes_per_node(nodes, edges, 831, False) < per      # fp32 rows are smaller than the image
    assert len(R.WindowedPages.layout(nodes, edges, 831, 1e9, True)) > len(a)   # half the budget: more windows


def _stream(ranges, B=10, passes=3, seed=7, rank=0):
    return R.WindowStream(ranges, B, passes, seed, rank)


def test_window_stream_steps_are_batches_of_distinct_pages_of_one_window():
    ranges = [(0, 95), (95, 140), (140, 260)]
    s = _stream(ranges)
    chunks = s.take(200)
    assert sum(len(steps) for _, steps in chunks) == 200
    for w, steps in chunks:
        p0, p1 = ranges[w]
        for ids in steps:
            assert ids.shape == (10,) and len(np.unique(ids)) == 10 and ids.min() >= 0 and ids.max() < p1 - p0
            assert np.all(np.diff(ids) > 0)                           # sorted: the resident rows of a batch ascend
    # consecutive chunks name different windows (a chunk is one visit, or the part of it this call covers)
    assert all(a[0] != b[0] for a, b in zip(chunks, chunks[1:]))


def test_window_stream_visits_every_page_passes_times_per_visit():
    ranges = [(0, 95), (95, 140), (140, 260)]
    s = _stream(ranges, B=10, passes=3)
    per_visit = {0: 9 * 3, 1: 4 * 3, 2: 12 * 3}                      # (pages // B) steps per pass, three passes
    seen = {}
    for w, steps in s.take(sum(per_visit.values())):                 # exactly one sweep over the three windows
        assert len(steps) == per_visit[w]
        cnt = np.bincount(np.concatenate(steps), minlength=ranges[w][1] - ranges[w][0])
        assert cnt.max() <= 3 and cnt.sum() == 10 * per_visit[w]      # a page at most once per pass; the pass's tail is dropped
        seen[w] = True
    assert set(seen) == {0, 1, 2}


def test_window_stream_is_deterministic_and_independent_of_how_it_is_consumed():
    ranges = [(0, 95), (95, 140), (140, 260), (260, 265)]            # (the last window holds fewer pages than a batch: skipped)
    flat = lambda chunks: [(w, tuple(ids)) for w, steps in chunks for ids in steps]
    a = flat(_stream(ranges).take(150))
    s2 = _stream(ranges)
    b = flat(s2.take(37)) + flat(s2.take(1)) + flat(s2.take(112))
    assert a == b
    assert all(w != 3 for w, _ in a)
    assert a != flat(_stream(ranges, rank=1).take(150))               # another rank draws another order
    assert a != flat(_stream(ranges, seed=8).take(150))


def test_window_stream_names_the_window_to_upload_next():
    ranges = [(0, 50), (50, 100), (100, 150)]
    s = _stream(ranges, B=10, passes=2)
    for _ in range(12):
        cur, nxt = s.peek_window(), s.next_window()
        chunk = s.take(10)                                           # one visit: 5 steps per pass, two passes
        assert len(chunk) == 1 and chunk[0][0] == cur
        assert s.peek_window() == nxt                                 # what was announced is what comes


def test_window_stream_refuses_a_batch_no_window_can_hold():
    with pytest.raises(ValueError):
        R.WindowStream([(0, 5), (5, 9)], 10, 2, 0)


def test_window_ranges_leave_no_undersized_last_window():
    """Filling every window to the brim left a last range of whatever remained -- possibly fewer pages than one batch, which the
    stream then never trained on (round-4 advisor).  Balanced ranges: 1 001 equal pages in windows of <= 100 are 11 windows of 91."""
    rs = R.window_ranges(np.full(1001, 200), 1.0, 100 * 200)
    assert len(rs) == 11 and min(p1 - p0 for p0, p1 in rs) >= 90
    assert _stream(rs, B=50).never_visited() == []
    assert _stream([(0, 95), (95, 99)], B=10).never_visited() == [95, 96, 97, 98]


def test_window_stream_skip_lands_where_take_would():
    """A resumed run fast-forwards the stream by the steps the interrupted run took: skip(n) == discarding take(n)."""
    ranges = [(0, 95), (95, 140), (140, 260), (260, 263)]           # (the last window holds fewer pages than a batch: skipped)
    for n in (0, 1, 8, 9, 27, 28, 100, 1234):
        a, b = _stream(ranges), _stream(ranges)
        a.take(n) if n else None
        b.skip(n)
        assert (a.sweep, a.pos, a.pas, a.off) == (b.sweep, b.pos, b.pas, b.off), n
        ta, tb = a.take(40), b.take(40)
        assert [w for w, _ in ta] == [w for w, _ in tb]
        for (_, sa), (_, sb) in zip(ta, tb):
            assert all((x == y).all() for x, y in zip(sa, sb))


def test_default_budget_keeps_a_fitting_set_resident_and_windows_a_larger_one():
    """A function of the device's TOTAL memory (round 5 derived it from the free memory at launch: the tier, the windows and with them
    the data order of a default run then depended on what other processes held at that moment)."""
    GB = 1e9
    assert R.default_budget_bytes(100 * GB, 288 * GB) is None                    # fits in half of the HBM: all resident
    assert R.default_budget_bytes(144 * GB, 288 * GB) is None
    assert R.default_budget_bytes(150 * GB, 288 * GB) == 0.45 * 288 * GB
    assert R.default_budget_bytes(500 * GB, 288 * GB) == 0.45 * 288 * GB         # PubLayNet's full train split at F0 = 831


def test_a_resumed_run_must_stand_on_the_interrupted_runs_layout():
    """The checkpoint records the residency layout (budget, tier, ranks, passes, window ranges of every rank); resuming under another one
    -- where the page stream's position means something else -- fails loudly instead of training on a different stream."""
    from gnn_tableextraction_amd.models.model_train import _check_layout
    lay = {'budget_gb': 12.0, 'tier': 'windowed', 'world': 2, 'passes': 8, 'batch_size': 100, 'ranges': [[[0, 500], [500, 900]], [[0, 450], [450, 900]]]}
    _check_layout(dict(lay), dict(lay), "ck")                                    # the same layout: fine
    _check_layout(dict(lay, budget_gb=11.9), dict(lay), "ck")                    # (the budget itself may differ while it yields the same windows)
    for change in ({'tier': 'owned'}, {'world': 4}, {'passes': 1}, {'ranges': [[[0, 900]], [[0, 900]]]}, {'batch_size': 50}):
        with pytest.raises(RuntimeError, match="another residency layout"):
            _check_layout(dict(lay), dict(lay, **change), "ck")
    # a checkpoint from before round 6 (no layout) resumes onto the all-resident tier only
    _check_layout(None, {'tier': 'all', 'world': 1, 'passes': None, 'batch_size': 100, 'ranges': None, 'budget_gb': 0.0}, "ck")
    with pytest.raises(RuntimeError, match="records no residency layout"):
        _check_layout(None, dict(lay), "ck")
