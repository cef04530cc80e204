"""Single-scale test iterator, contract of lib/iterators/MNIteratorTest.py:6-66: batches of whole images resized to
`test_scale`, data = [data (B,3,H,W), im_info (B,3) = resized h, w, scale, im_ids (B,)]."""
import numpy as np
import torch

import sniper_amd.mx as mx

from .. import hip
from .MNIteratorBase import MNIteratorBase, WholeImageTestMixin, orientation_order


class MNIteratorTest(WholeImageTestMixin, MNIteratorBase):
    data_name = ['data', 'im_info', 'im_ids']

    def __init__(self, roidb, config, test_scale, batch_size=4, threads=8, nGPUs=1, pad_rois_to=400, crop_size=(512, 512),
                 num_classes=None, image_cache=None):
        self._init_test(roidb, config, test_scale, crop_size, num_classes, image_cache)
        MNIteratorBase.__init__(self, roidb, config, batch_size, threads, nGPUs, pad_rois_to, True)
        self.reset()

    def reset(self):          # roidb order, no shuffle; a short group repeats its first image (:52-65)
        self._set_order(orientation_order([r['width'] for r in self.roidb], [r['height'] for r in self.roidb], self.batch_size, 'first'))

    def _get_batch(self, roidb):
        landscape = roidb[0]['width'] >= roidb[0]['height']
        canvas = list(self.test_scale[:2]) if landscape else [self.test_scale[1], self.test_scale[0]]
        images = torch.empty((len(roidb), 3, canvas[0], canvas[1]), dtype=torch.float32, device=hip.require_gpu())
        info = np.zeros((len(roidb), 3), np.float32)
        for i, r in enumerate(roidb):
            scale, (h, w) = self.im_worker.worker([r['image'], canvas, r['flipped']], images[i])
            info[i] = (h, w, scale)
        ids = np.asarray(self._current_units(), np.float32)
        return self._emit([mx.nd.NDArray(images), mx.nd.array(info), mx.nd.array(ids)])
