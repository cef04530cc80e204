"""Iterator base with the contract of lib/iterators/MNIteratorBase.py:6-111 (aspect-ratio grouping, provide_* and
batch bookkeeping).  The reference's ThreadPool / ThreadPoolExecutor for image loading do not exist here: the image
path is one GPU kernel per image (sniper_amd/data/im_worker.py)."""
import numpy as np

import sniper_amd.mx as mx


class MNIteratorBase(mx.io.DataIter):
    def __init__(self, roidb, config, batch_size, threads, nGPUs, pad_rois_to, single_size_change):
        super(MNIteratorBase, self).__init__()
        assert batch_size % nGPUs == 0, 'batch_size should be divisible by number of GPUs'
        self.cur_i = 0
        self.roidb = roidb
        self.batch_size = batch_size
        self.pixel_mean = config.network.PIXEL_MEANS
        self.n_per_gpu = batch_size // nGPUs
        self.batch = None
        self.cfg = config
        self.n_expected_roi = pad_rois_to
        self.single_size_change = single_size_change
        self.reset()
        self.get_batch()

    def get_batch_size(self):
        return self.batch_size

    def __len__(self):
        return len(self.inds)

    @property
    def provide_data(self):
        return [(k, tuple(v.shape)) for k, v in zip(self.data_name, self.data)]

    @property
    def provide_label(self):
        if self.label_name:
            return [(k, tuple(v.shape)) for k, v in zip(self.label_name, self.label)]
        return None

    provide_data_single = provide_data
    provide_label_single = provide_label

    def reset(self):
        self.cur_i = 0
        widths = np.array([r['width'] for r in self.roidb])
        heights = np.array([r['height'] for r in self.roidb])
        horz_inds = np.where(widths >= heights)[0]
        vert_inds = np.where(widths < heights)[0]
        if horz_inds.shape[0] % self.batch_size > 0:
            extra = self.batch_size - (horz_inds.shape[0] % self.batch_size)
            horz_inds = np.hstack((horz_inds, horz_inds[0:extra]))
        if vert_inds.shape[0] % self.batch_size > 0:
            extra = self.batch_size - (vert_inds.shape[0] % self.batch_size)
            vert_inds = np.hstack((vert_inds, vert_inds[0:extra]))
        inds = np.hstack((np.random.permutation(horz_inds), np.random.permutation(vert_inds)))
        assert inds.shape[0] % self.batch_size == 0, 'The number of samples here should be divisible by batch size'
        if not self.single_size_change:
            inds_ = np.reshape(inds, (-1, self.batch_size))
            inds = np.reshape(inds_[np.random.permutation(np.arange(inds_.shape[0])), :], (-1,))
        self.inds = inds
        self.size = len(self.inds)

    def iter_next(self):
        return self.get_batch()

    def next(self):
        if self.iter_next():
            return self.batch
        raise StopIteration

    __next__ = next

    def get_batch(self):
        if self.cur_i >= self.size:
            return False
        cur_roidbs = [self.roidb[self.inds[i % self.size]] for i in range(self.cur_i, self.cur_i + self.batch_size)]
        self.batch = self._get_batch(cur_roidbs)
        self.cur_i += self.batch_size
        return True

    def get_index(self):
        return self.cur_i // self.batch_size

    def getindex(self):
        return self.get_index()

    def getpad(self):
        return 0

    def _get_batch(self, roidb):
        raise NotImplementedError('This method should be implemented in the inherited classes')
