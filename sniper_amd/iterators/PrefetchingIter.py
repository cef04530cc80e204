"""Background prefetch around a single iterator, contract of lib/iterators/PrefetchingIter.py:15-148 (used by main_train.py:142
and Tester.__init__, lib/inference.py:33-34): the reference keeps one batch ahead; here two by default (a batch of 20 chips is
63 MB of a 288 GB card, and one spare batch absorbs a burst of image decodes -- the order of the batches is the iterator's, unchanged),
`depth` batches ahead for a consumer that keeps several batches in flight (the lanes of Tester.get_detections).  The worker thread binds the iterator's GPU: batch
assembly here is kernel launches (anchor labelling, image preparation), enqueued while the main thread is busy with the previous
step -- on the worker's own HIP stream.  A batch carries `ready_event`, recorded on that stream behind those launches: the consumer
waits for it on the device (`adopt_batch`: stream.wait_event + record_stream) -- the host never blocks on the assembly."""
import collections
import threading

import numpy as np

import sniper_amd.mx as mx


def stamp_ready(batch, ev):
    """Put the assembly stream's event on the batch AND on each of its device arrays: a reader that is not the adopting consumer
    (reference / debug code calling `batch.data[i].asnumpy()`, a metric reading a label) waits for it in NDArray.asnumpy instead of
    copying a half-written buffer -- `.cpu()` orders only with the CURRENT stream."""
    batch.ready_event = ev
    for group in (getattr(batch, 'data', None), getattr(batch, 'label', None)):
        for a in (group or ()):
            if hasattr(a, '_data') and not isinstance(getattr(a, '_data', None), np.ndarray):
                try:
                    a._ready_event = ev
                except AttributeError:
                    pass


def adopt_batch(batch, stream=None):
    """Consumer side of a batch assembled on the worker's stream: make `stream` (default: the current one) wait for the batch's
    `ready_event` and tell the allocator that the batch's device tensors are in use on it (`record_stream`) -- the host runs
    ahead of the device, so a batch object is dropped while the copy that reads it is still queued, and without the record the
    worker's next batch could be written into the same memory first."""
    ev = getattr(batch, 'ready_event', None)
    if ev is None:
        return
    import torch
    st = stream or torch.cuda.current_stream()
    st.wait_event(ev)
    for a in list(batch.data or []) + list(batch.label or []):
        t = getattr(a, '_data', a)
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(st)


class PrefetchingIter(mx.io.DataIter):
    def __init__(self, iters, rename_data=None, rename_label=None, depth=2, own_stream=True):
        super(PrefetchingIter, self).__init__()
        if not isinstance(iters, list):
            iters = [iters]
        self.n_iter = len(iters)
        assert self.n_iter == 1, 'Our prefetching iter only support 1 DataIter'
        self.iters = iters
        self.rename_data, self.rename_label = rename_data, rename_label
        self.batch_size = self.iters[0].get_batch_size() if hasattr(self.iters[0], 'get_batch_size') else \
            self.provide_data[0][1][0]
        self.depth = max(1, int(depth))
        # own_stream: the worker assembles batches on a HIP stream of its own (training: the consumer's stream is busy with the step).
        # False: on the consumer's default stream -- the test-time Tester, whose lanes already hold the runtime's four hardware
        # queues: a fifth stream shares a queue with a lane and the 64-image pass measured 310 instead of 262 ms
        # (profiles/r05_infer_prefetch_stream_ab.txt)
        self.own_stream = bool(own_stream)
        self._cv = threading.Condition()
        self._queue = collections.deque()       # (batch or None at the end of the epoch, exception or None)
        self._epoch = 0                         # bumped by reset(): a batch assembled across a reset is dropped
        self._exhausted = False
        self._busy = False
        self.started = True
        self.current_batch = None
        try:
            import torch
            self._device = torch.cuda.current_device() if torch.cuda.is_available() else None
        except Exception:  # noqa: BLE001
            self._device = None

        self._start()

    def _prefetch_loop(self):
        """The worker thread's body: a METHOD, not a closure kept on self -- a closure over self stored on self is a reference
        cycle that outlives close() until a cyclic collection sees it (and forever once the heap is frozen, engine/executor.py::
        settle_heap); a finished Thread drops its target, so a closed iterator dies by reference counting."""
        stream = None
        if self._device is not None:
            import torch
            torch.cuda.set_device(self._device)
            # the worker's OWN stream: batch assembly is small uploads from pageable memory (synchronous copies: each waits for
            # everything queued before it on its stream) and small kernels -- on the consumer's stream every one of them would wait
            # for the training step in flight, and a batch took longer to assemble than a step to run (profiles/r05_fit_path.txt)
            import os
            if self.own_stream and os.environ.get('SNIPER_PREFETCH_STREAM', '1') != '0':      # (=0: never, A/B)
                stream = self._stream = torch.cuda.Stream(device=self._device)
                torch.cuda.set_stream(stream)
        while True:
            with self._cv:
                while self.started and (self._exhausted or len(self._queue) >= self.depth):
                    self._cv.wait()
                if not self.started:
                    break
                epoch, self._busy = self._epoch, True
            batch, error = None, None
            try:
                batch = self.iters[0].next()
                if self._device is not None:
                    import torch
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    stamp_ready(batch, ev)          # the consumer: stream.wait_event + adopt_batch (record_stream), below
            except StopIteration:
                batch = None
            except Exception as e:  # noqa: BLE001 -- surface worker failures in the consumer thread
                batch, error = None, e
            with self._cv:
                self._busy = False
                if epoch == self._epoch:
                    self._queue.append((batch, error))
                    self._exhausted = batch is None
                self._cv.notify_all()

    def _start(self):
        self.started = True
        self.prefetch_thread = threading.Thread(target=self._prefetch_loop, daemon=True)
        self.prefetch_thread.start()

    def close(self):
        """End the worker thread (it otherwise waits for a reset() for as long as the process lives).  Not final: reset()
        starts a new worker, so an iterator a Tester closed after one scale serves the next `set_scale` (demo.py reuses one
        Tester across scales)."""
        self.__del__()
        t = getattr(self, 'prefetch_thread', None)
        if t is not None and t is not threading.current_thread():
            t.join(timeout=30)

    def __del__(self):
        self.started = False
        try:
            with self._cv:
                self._cv.notify_all()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def get_batch_size(self):
        if not hasattr(self.iters[0], 'get_batch_size'):
            raise NotImplementedError
        return self.iters[0].get_batch_size()

    def __len__(self):
        return len(self.iters[0])

    def _renamed(self, descs, maps):
        descs = list(descs or [])
        if maps is None:
            return descs
        return [(maps[0].get(d[0], d[0]),) + tuple(d[1:]) for d in descs]

    @property
    def provide_data(self):
        return self._renamed(self.iters[0].provide_data, self.rename_data)

    @property
    def provide_label(self):
        return self._renamed(self.iters[0].provide_label, self.rename_label)

    @property
    def provide_data_single(self):
        return self.iters[0].provide_data_single

    @property
    def provide_label_single(self):
        return self.iters[0].provide_label_single

    def reset(self):
        with self._cv:
            while self._busy:                   # the worker is inside iters[0].next(): let it finish, then drop what it made
                self._cv.wait()
            self._queue.clear()
            self._epoch += 1
            self.iters[0].reset()
            self._exhausted = False
            self._cv.notify_all()
        if not self.started or not self.prefetch_thread.is_alive():          # closed earlier: a fresh worker for the new epoch
            self._start()

    def iter_next(self):
        with self._cv:
            while not self._queue:
                self._cv.wait()
            batch, error = self._queue[0]
            if batch is not None:
                self._queue.popleft()           # (the end-of-epoch marker stays: every further call answers False until reset)
                self._cv.notify_all()
            elif error is not None:
                self._queue[0] = (None, None)
        if error is not None:
            raise error
        if batch is None:
            return False
        self.current_batch = batch
        return True

    def next(self):
        if self.iter_next():
            return self.current_batch
        raise StopIteration

    __next__ = next

    def __iter__(self):
        return self

    def getdata(self):
        return self.current_batch.data

    def getlabel(self):
        return self.current_batch.label

    def getindex(self):
        return self.current_batch.index

    def getpad(self):
        return self.current_batch.pad
