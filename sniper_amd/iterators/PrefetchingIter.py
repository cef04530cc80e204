"""One-deep background prefetch around a single iterator, contract of lib/iterators/PrefetchingIter.py:15-148
(used by main_train.py:142 and Tester.__init__, lib/inference.py:33-34).  The worker thread binds the iterator's GPU:
batch assembly here is kernel launches (anchor labelling, image preparation), enqueued while the main thread is
busy with the previous step; both threads use the device's default stream, so ordering needs no extra events."""
import threading

import sniper_amd.mx as mx


class PrefetchingIter(mx.io.DataIter):
    def __init__(self, iters, rename_data=None, rename_label=None):
        super(PrefetchingIter, self).__init__()
        if not isinstance(iters, list):
            iters = [iters]
        self.n_iter = len(iters)
        assert self.n_iter == 1, 'Our prefetching iter only support 1 DataIter'
        self.iters = iters
        self.rename_data, self.rename_label = rename_data, rename_label
        self.batch_size = self.iters[0].get_batch_size() if hasattr(self.iters[0], 'get_batch_size') else \
            self.provide_data[0][1][0]
        self.data_ready = threading.Event()
        self.data_taken = threading.Event()
        self.data_taken.set()
        self.started = True
        self.current_batch = None
        self.next_batch = None
        self.error = None
        try:
            import torch
            self._device = torch.cuda.current_device() if torch.cuda.is_available() else None
        except Exception:  # noqa: BLE001
            self._device = None

        def prefetch_func():
            if self._device is not None:
                import torch
                torch.cuda.set_device(self._device)
            while True:
                self.data_taken.wait()
                if not self.started:
                    break
                try:
                    self.next_batch = self.iters[0].next()
                except StopIteration:
                    self.next_batch = None
                except Exception as e:  # noqa: BLE001 -- surface worker failures in the consumer thread
                    self.next_batch, self.error = None, e
                self.data_taken.clear()
                self.data_ready.set()
        self.prefetch_thread = threading.Thread(target=prefetch_func, daemon=True)
        self.prefetch_thread.start()

    def __del__(self):
        self.started = False
        self.data_taken.set()

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
        self.data_ready.wait()
        self.iters[0].reset()
        self.data_ready.clear()
        self.data_taken.set()

    def iter_next(self):
        self.data_ready.wait()
        if self.error is not None:
            e, self.error = self.error, None
            raise e
        if self.next_batch is None:
            return False
        self.current_batch = self.next_batch
        self.data_ready.clear()
        self.data_taken.set()
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
