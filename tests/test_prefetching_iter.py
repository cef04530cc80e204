"""PrefetchingIter (lib/iterators/PrefetchingIter.py:15-148): order, end of epoch, reset mid-epoch, worker errors, and the
deeper queue the test-time lanes use -- with a plain Python iterator, no GPU."""
import threading
import time

import pytest

from sniper_amd.iterators.PrefetchingIter import PrefetchingIter


class _Batch(object):
    def __init__(self, k):
        self.k, self.data, self.label, self.pad, self.index = k, [k], None, 0, k


class _Counter(object):
    """Hands out batches 0..n-1; records how far ahead of the consumer it was asked to run."""
    provide_data, provide_label = [('data', (2, 3))], None
    provide_data_single, provide_label_single = provide_data, provide_label

    def __init__(self, n, fail_at=None):
        self.n, self.fail_at, self.i, self.made = n, fail_at, 0, 0

    def get_batch_size(self):
        return 2

    def __len__(self):
        return self.n

    def reset(self):
        self.i = 0

    def next(self):
        if self.i >= self.n:
            raise StopIteration
        if self.i == self.fail_at:
            self.i += 1
            raise ValueError('boom')
        self.i += 1
        self.made += 1
        return _Batch(self.i - 1)


def _wait_for(cond, seconds=5.0):
    t0 = time.time()
    while not cond() and time.time() - t0 < seconds:
        time.sleep(0.005)
    return cond()


@pytest.mark.parametrize('depth', [1, 3])
def test_order_end_of_epoch_and_reset(depth):
    src = _Counter(7)
    it = PrefetchingIter(src, depth=depth)
    assert [b.k for b in it] == list(range(7))
    assert not it.iter_next() and not it.iter_next()         # stays finished until reset
    it.reset()
    got = [it.next().k for _ in range(3)]
    it.reset()                                                # mid-epoch: whatever was prefetched is dropped
    assert got == [0, 1, 2] and [b.k for b in it] == list(range(7))
    it.close()
    assert _wait_for(lambda: not it.prefetch_thread.is_alive())


@pytest.mark.parametrize('depth', [1, 4])
def test_runs_exactly_depth_batches_ahead(depth):
    src = _Counter(20)
    it = PrefetchingIter(src, depth=depth)
    assert _wait_for(lambda: src.made == depth)
    time.sleep(0.05)
    assert src.made == depth                                  # the queue is full: the worker waits
    assert it.next().k == 0
    assert _wait_for(lambda: src.made == depth + 1)
    it.close()


def test_worker_error_surfaces_in_the_consumer():
    it = PrefetchingIter(_Counter(5, fail_at=2), depth=2)
    assert it.next().k == 0 and it.next().k == 1
    with pytest.raises(ValueError):
        it.next()
    assert not it.iter_next()
    it.reset()
    assert it.next().k == 0
    it.close()
    assert threading.active_count() < 50


def test_reset_after_close_starts_a_new_worker():
    """Tester.get_detections closes the iterator it wrapped; the reference reuses one Tester across scales (demo.py:
    `tester.set_scale(s)` -> `PrefetchingIter.reset()` -> next `get_detections`).  A reset after close must serve a full epoch
    again instead of waiting on a queue nobody fills."""
    src = _Counter(5)
    it = PrefetchingIter(src, depth=2)
    assert [b.k for b in it] == list(range(5))
    it.close()
    assert not it.prefetch_thread.is_alive()
    assert not it.iter_next()                                  # closed and exhausted: still answers, does not hang
    for _ in range(2):
        it.reset()
        assert [b.k for b in it] == list(range(5))
        it.close()
    assert _wait_for(lambda: not it.prefetch_thread.is_alive())
