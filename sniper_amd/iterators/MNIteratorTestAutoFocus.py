"""AutoFocus test iterator, contract of lib/iterators/MNIteratorTestAutoFocus.py:13-141: the units are the
`inference_crops` (FocusChips) of every image, sorted by area and grouped by orientation so that a batch pads little;
data = [data (B,3,Hm,Wm), im_info (B,3), im_ids (B,), chip_ids (B,)] with (Hm, Wm) the largest resized chip of the batch."""
import math

import numpy as np
import torch

import sniper_amd.mx as mx

from .. import hip
from ..data.im_worker import im_worker, target_scale
from .MNIteratorBase import MNIteratorBase


class MNIteratorTestAutoFocus(MNIteratorBase):
    def __init__(self, roidb, config, test_scale, batch_size=4, threads=8, nGPUs=1, pad_rois_to=400, crop_size=(512, 512),
                 num_classes=None, image_cache=None):
        self.crop_size = crop_size
        self.image_cache = image_cache         # data/im_worker.py::DeviceImageCache shared by the scales of one pass (or None)
        self.num_classes = num_classes if num_classes else roidb[0]['gt_overlaps'].shape[1]
        self.data_name = ['data', 'im_info', 'im_ids', 'chip_ids']
        self.label_name = None
        self.label = []
        self.context_size = 320
        self.im_worker = im_worker(crop_size=None if not self.crop_size else self.crop_size[0], cfg=config,
                                   target_size=test_scale, image_cache=image_cache)
        self.test_scale = test_scale
        # the base constructor assembles one batch only to learn the shapes provide_data reports (MNIteratorBase.py:23-24):
        # that one allocates the tensors and skips the image preparation (a pass re-creates this iterator per scale)
        self._shapes_only = True
        super(MNIteratorTestAutoFocus, self).__init__(roidb, config, batch_size, threads, nGPUs, pad_rois_to, True)
        self._shapes_only = False
        self.reset()

    def set_scale(self, scale):
        self.test_scale = scale
        self.im_worker = im_worker(crop_size=None if not self.crop_size else self.crop_size[0], cfg=self.cfg, target_size=scale,
                                   image_cache=self.image_cache)

    def _get_batch(self, roidb, chip_ids, im_ids):
        n_batch = len(roidb)
        max_size = [0, 0]
        chips, scales = [], []
        local_chip_ids = np.zeros(n_batch, np.float32)
        for i, r in enumerate(roidb):
            scale = target_scale(r['width'], r['height'], self.test_scale)
            cchip_id = r['crop_mapping'][chip_ids[i]]
            cur_chip = r['inference_crops'][cchip_id]
            local_chip_ids[i] = cchip_id
            max_size[0] = max(max_size[0], int(math.ceil((cur_chip[3] - cur_chip[1]) * scale)))
            max_size[1] = max(max_size[1], int(math.ceil((cur_chip[2] - cur_chip[0]) * scale)))
            chips.append(cur_chip)
            scales.append(scale)
        im_tensor = torch.empty((n_batch, 3, max_size[0], max_size[1]), dtype=torch.float32, device=hip.require_gpu())
        im_info = np.zeros((n_batch, 3), np.float32)
        for i in range(n_batch if not self._shapes_only else 0):
            scale, (h, w) = self.im_worker.worker_autofocus([roidb[i]['image'], max_size, roidb[i]['flipped'], chips[i], scales[i]],
                                                            im_tensor[i])
            im_info[i] = [h, w, scale]
        self.data = [mx.nd.NDArray(im_tensor), mx.nd.array(im_info), mx.nd.array(np.asarray(im_ids, np.float32)),
                     mx.nd.array(local_chip_ids)]
        return mx.io.DataBatch(data=self.data, label=self.label, pad=self.getpad(), index=self.getindex(),
                               provide_data=self.provide_data, provide_label=self.provide_label)

    def get_batch(self):
        if self.cur_i >= self.size:
            return False
        cur_chip_ids = [self.inds[i % self.size] for i in range(self.cur_i, self.cur_i + self.batch_size)]
        cur_roidb_ids = [self.crop2im[i] for i in cur_chip_ids]
        cur_roidbs = [self.roidb[i] for i in cur_roidb_ids]
        self.batch = self._get_batch(cur_roidbs, cur_chip_ids, cur_roidb_ids)
        self.cur_i += self.batch_size
        return True

    def reset(self):
        self.cur_i = 0
        self.crop2im = {}
        sizes = []
        crop_counter = 0
        for i, r in enumerate(self.roidb):
            local_crop_mapping = {}
            for local_counter, crop in enumerate(r['inference_crops']):
                sizes.append([crop[2] - crop[0], crop[3] - crop[1]])
                self.crop2im[crop_counter] = i
                local_crop_mapping[crop_counter] = local_counter
                crop_counter += 1
            r['crop_mapping'] = local_crop_mapping
        sizes = np.array(sizes, dtype=np.float64).reshape(-1, 2)
        self.inds = (sizes[:, 0] * sizes[:, 1]).argsort()            # sort by area (:106-109)
        widths, heights = sizes[self.inds, 0], sizes[self.inds, 1]
        horz_inds = np.where(widths >= heights)[0]
        vert_inds = np.where(widths < heights)[0]
        if horz_inds.shape[0] % self.batch_size > 0:
            extra = self.batch_size - (horz_inds.shape[0] % self.batch_size)
            horz_inds = np.hstack((horz_inds, horz_inds[-extra:]))
        if vert_inds.shape[0] % self.batch_size > 0:
            extra = self.batch_size - (vert_inds.shape[0] % self.batch_size)
            vert_inds = np.hstack((vert_inds, vert_inds[-extra:]))
        inds = np.hstack((horz_inds, vert_inds)).astype(int)
        if inds.shape[0] % self.batch_size > 0:
            extra = self.batch_size - (inds.shape[0] % self.batch_size)
            inds = np.hstack((inds, inds[-extra:]))
        self.inds = self.inds[inds]
        assert self.inds.shape[0] % self.batch_size == 0, 'The number of samples here should be divisible by batch size'
        self.size = len(self.inds)
