"""Iterator base, contract of lib/iterators/MNIteratorBase.py:6-111: the epoch order groups landscape and portrait units into whole
batches (a batch never mixes orientations, so it pads little), `provide_data` / `provide_label` describe the current batch, and
`next()` hands out what `_get_batch` of the subclass assembled.  The reference's ThreadPool / ThreadPoolExecutor for image loading do
not exist here: the image path is one GPU kernel per image (sniper_amd/data/im_worker.py).

`orientation_order` is the one statement of the grouping the three iterators share; what differs between them is how a group is
filled up to a whole number of batches (`fill`) and whether numpy shuffles it:
    'head'  training (MNIteratorBase.py:66-82): the group's first items once more, then a permutation of each group, then -- unless
            the scale may only change once per epoch -- a permutation of the batches; three draws from numpy's global generator, in
            the reference's order, so a seeded run visits the same images;
    'first' single-scale test (MNIteratorTest.py:52-65): the group's first item repeated;
    'tail'  AutoFocus test (MNIteratorTestAutoFocus.py:119-133): the group's last items once more.
"""
import numpy as np

import sniper_amd.mx as mx


def _fill_group(idx, batch_size, fill):
    short = (-idx.shape[0]) % batch_size
    if short == 0:
        return idx
    if fill == 'head':
        extra = idx[:short]
    elif fill == 'first':
        extra = idx[np.zeros(short, dtype=int)]
    else:
        extra = idx[-short:]
    return np.concatenate((idx, extra))


def orientation_order(widths, heights, batch_size, fill, shuffle=False, shuffle_batches=False):
    """Indices of the units, landscape group first, every group a whole number of batches."""
    widths, heights = np.asarray(widths), np.asarray(heights)
    groups = [_fill_group(np.where(sel)[0], batch_size, fill) for sel in (widths >= heights, widths < heights)]
    if shuffle:
        groups = [np.random.permutation(g) for g in groups]
    order = np.concatenate(groups).astype(int)
    if fill == 'tail' and order.shape[0] % batch_size:       # (a group shorter than its own shortfall: the whole order is filled once more, :128-131)
        order = np.concatenate((order, order[-((-order.shape[0]) % batch_size):]))
    assert order.shape[0] % batch_size == 0, 'The number of samples here should be divisible by batch size'
    if shuffle_batches:
        by_batch = order.reshape(-1, batch_size)
        order = by_batch[np.random.permutation(np.arange(by_batch.shape[0]))].reshape(-1)
    return order


def _describe(names, arrays):
    return [(n, tuple(a.shape)) for n, a in zip(names, arrays)]


class MNIteratorBase(mx.io.DataIter):
    def __init__(self, roidb, config, batch_size, threads, nGPUs, pad_rois_to, single_size_change):
        super(MNIteratorBase, self).__init__()
        assert batch_size % nGPUs == 0, 'batch_size should be divisible by number of GPUs'
        self.roidb, self.cfg = roidb, config
        self.batch_size, self.n_per_gpu = batch_size, batch_size // nGPUs
        self.pixel_mean = config.network.PIXEL_MEANS
        self.n_expected_roi = pad_rois_to
        self.single_size_change = single_size_change
        self.batch, self.cur_i = None, 0
        self.reset()
        self.get_batch()        # (the reference does: provide_data is asked for before the first next())

    # ---- what the Module asks
    provide_data = property(lambda self: _describe(self.data_name, self.data))
    provide_label = property(lambda self: _describe(self.label_name, self.label) if self.label_name else None)
    provide_data_single, provide_label_single = provide_data, provide_label

    def get_batch_size(self):
        return self.batch_size

    def __len__(self):
        return len(self.inds)

    def get_index(self):
        return self.cur_i // self.batch_size

    getindex = get_index

    def getpad(self):
        return 0

    # ---- the epoch
    def _set_order(self, order):
        self.cur_i, self.inds, self.size = 0, order, len(order)

    def reset(self):
        self._set_order(orientation_order([r['width'] for r in self.roidb], [r['height'] for r in self.roidb], self.batch_size,
                                          'head', shuffle=True, shuffle_batches=not self.single_size_change))

    def _current_units(self):
        return [self.inds[i % self.size] for i in range(self.cur_i, self.cur_i + self.batch_size)]

    def get_batch(self):
        if self.cur_i >= self.size:
            return False
        self.batch = self._get_batch([self.roidb[u] for u in self._current_units()])
        self.cur_i += self.batch_size
        return True

    def iter_next(self):
        return self.get_batch()

    def next(self):
        if not self.iter_next():
            raise StopIteration
        return self.batch

    __next__ = next

    def _get_batch(self, roidb):
        raise NotImplementedError('This method should be implemented in the inherited classes')


class WholeImageTestMixin(object):
    """What the two test iterators share: names, the per-scale im_worker, and the (B,3,H,W) / (B,3) batch tensors."""
    label_name, label, context_size = None, [], 320

    def _init_test(self, roidb, config, test_scale, crop_size, num_classes, image_cache):
        self.crop_size, self.image_cache = crop_size, image_cache      # image_cache: data/im_worker.py::DeviceImageCache of one pass (or None)
        self.num_classes = num_classes if num_classes else roidb[0]['gt_overlaps'].shape[1]
        self._cfg_for_worker = config
        self.set_scale(test_scale)

    def set_scale(self, scale):
        from ..data.im_worker import im_worker
        self.test_scale = scale
        self.im_worker = im_worker(crop_size=self.crop_size[0] if self.crop_size else None, cfg=self._cfg_for_worker, target_size=scale,
                                   image_cache=self.image_cache)

    def _emit(self, arrays):
        self.data = arrays
        return mx.io.DataBatch(data=self.data, label=self.label, pad=self.getpad(), index=self.getindex(),
                               provide_data=self.provide_data, provide_label=self.provide_label)
