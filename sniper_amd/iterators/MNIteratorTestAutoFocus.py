"""AutoFocus test iterator, contract of lib/iterators/MNIteratorTestAutoFocus.py:13-141: the units are the
`inference_crops` (FocusChips) of every image, sorted by area and grouped by orientation so that a batch pads little;
data = [data (B,3,Hm,Wm), im_info (B,3), im_ids (B,), chip_ids (B,)] with (Hm, Wm) the largest resized chip of the batch."""
import math

import numpy as np
import torch

import sniper_amd.mx as mx

from .. import hip
from ..data.im_worker import target_scale
from .MNIteratorBase import MNIteratorBase, WholeImageTestMixin, orientation_order


class MNIteratorTestAutoFocus(WholeImageTestMixin, MNIteratorBase):
    data_name = ['data', 'im_info', 'im_ids', 'chip_ids']

    def __init__(self, roidb, config, test_scale, batch_size=4, threads=8, nGPUs=1, pad_rois_to=400, crop_size=(512, 512),
                 num_classes=None, image_cache=None):
        self._init_test(roidb, config, test_scale, crop_size, num_classes, image_cache)
        # the base constructor assembles one batch only to learn the shapes provide_data reports (MNIteratorBase.py:23-24):
        # that one allocates the tensors and skips the image preparation (a pass re-creates this iterator per scale)
        self._shapes_only = True
        MNIteratorBase.__init__(self, roidb, config, batch_size, threads, nGPUs, pad_rois_to, True)
        self._shapes_only = False
        self.reset()

    def reset(self):
        """Global chip numbering in roidb order (`crop2im`, each image's `crop_mapping`: global -> local), chips by ascending area
        (:106-109), then the orientation groups over THAT order, short groups filled with their last chips (:119-133)."""
        self.crop2im, extents = {}, []
        for im, r in enumerate(self.roidb):
            first = len(extents)
            extents.extend((c[2] - c[0], c[3] - c[1]) for c in r['inference_crops'])
            r['crop_mapping'] = {first + k: k for k in range(len(extents) - first)}
            self.crop2im.update((g, im) for g in r['crop_mapping'])
        extents = np.array(extents, dtype=np.float64).reshape(-1, 2)
        by_area = (extents[:, 0] * extents[:, 1]).argsort()
        grouped = orientation_order(extents[by_area, 0], extents[by_area, 1], self.batch_size, 'tail')
        self._set_order(by_area[grouped])

    def get_batch(self):
        if self.cur_i >= self.size:
            return False
        chips = self._current_units()
        images = [self.crop2im[c] for c in chips]
        self.batch = self._get_batch([self.roidb[i] for i in images], chips, images)
        self.cur_i += self.batch_size
        return True

    def _get_batch(self, roidb, chip_ids, im_ids):
        local = [r['crop_mapping'][g] for r, g in zip(roidb, chip_ids)]
        boxes = [r['inference_crops'][k] for r, k in zip(roidb, local)]
        scales = [target_scale(r['width'], r['height'], self.test_scale) for r in roidb]
        canvas = [max(int(math.ceil((b[3] - b[1]) * s)) for b, s in zip(boxes, scales)),
                  max(int(math.ceil((b[2] - b[0]) * s)) for b, s in zip(boxes, scales))]
        images = torch.empty((len(roidb), 3, canvas[0], canvas[1]), dtype=torch.float32, device=hip.require_gpu())
        info = np.zeros((len(roidb), 3), np.float32)
        if not self._shapes_only:
            for i, r in enumerate(roidb):
                scale, (h, w) = self.im_worker.worker_autofocus([r['image'], canvas, r['flipped'], boxes[i], scales[i]], images[i])
                info[i] = (h, w, scale)
        return self._emit([mx.nd.NDArray(images), mx.nd.array(info), mx.nd.array(np.asarray(im_ids, np.float32)),
                           mx.nd.array(np.asarray(local, np.float32))])
