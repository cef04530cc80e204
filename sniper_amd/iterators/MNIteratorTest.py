"""Single-scale test iterator, contract of lib/iterators/MNIteratorTest.py:6-66: batches of whole images resized to
`test_scale`, data = [data (B,3,H,W), im_info (B,3) = resized h, w, scale, im_ids (B,)]."""
import numpy as np
import torch

import sniper_amd.mx as mx

from .. import hip
from ..data.im_worker import im_worker
from .MNIteratorBase import MNIteratorBase


class MNIteratorTest(MNIteratorBase):
    def __init__(self, roidb, config, test_scale, batch_size=4, threads=8, nGPUs=1, pad_rois_to=400, crop_size=(512, 512),
                 num_classes=None, image_cache=None):
        self.crop_size = crop_size
        self.image_cache = image_cache         # data/im_worker.py::DeviceImageCache shared by the scales of one pass (or None)
        self.num_classes = num_classes if num_classes else roidb[0]['gt_overlaps'].shape[1]
        self.data_name = ['data', 'im_info', 'im_ids']
        self.label_name = None
        self.label = []
        self.context_size = 320
        self.im_worker = im_worker(crop_size=None if not self.crop_size else self.crop_size[0], cfg=config,
                                   target_size=test_scale, image_cache=image_cache)
        self.test_scale = test_scale
        super(MNIteratorTest, self).__init__(roidb, config, batch_size, threads, nGPUs, pad_rois_to, True)
        self.reset()

    def set_scale(self, scale):
        self.test_scale = scale
        self.im_worker = im_worker(crop_size=None if not self.crop_size else self.crop_size[0], cfg=self.cfg, target_size=scale,
                                   image_cache=self.image_cache)

    def _get_batch(self, roidb):
        n_batch = len(roidb)
        im_ids = np.array([self.inds[i % self.size] for i in range(self.cur_i, self.cur_i + self.batch_size)])
        hor_flag = roidb[0]['width'] >= roidb[0]['height']
        max_size = [self.test_scale[0], self.test_scale[1]] if hor_flag else [self.test_scale[1], self.test_scale[0]]
        im_tensor = torch.empty((n_batch, 3, max_size[0], max_size[1]), dtype=torch.float32, device=hip.require_gpu())
        im_info = np.zeros((n_batch, 3), np.float32)
        for i in range(n_batch):
            scale, (h, w) = self.im_worker.worker([roidb[i]['image'], max_size, roidb[i]['flipped']], im_tensor[i])
            im_info[i] = [h, w, scale]
        self.data = [mx.nd.NDArray(im_tensor), mx.nd.array(im_info), mx.nd.array(im_ids.astype(np.float32))]
        return mx.io.DataBatch(data=self.data, label=self.label, pad=self.getpad(), index=self.getindex(),
                               provide_data=self.provide_data, provide_label=self.provide_label)

    def reset(self):
        self.cur_i = 0
        widths = np.array([r['width'] for r in self.roidb])
        heights = np.array([r['height'] for r in self.roidb])
        horz_inds = np.where(widths >= heights)[0]
        vert_inds = np.where(widths < heights)[0]
        if horz_inds.shape[0] % self.batch_size > 0:
            extra = self.batch_size - (horz_inds.shape[0] % self.batch_size)
            horz_inds = np.hstack((horz_inds, horz_inds[np.zeros(extra, dtype='int')]))
        if vert_inds.shape[0] % self.batch_size > 0:
            extra = self.batch_size - (vert_inds.shape[0] % self.batch_size)
            vert_inds = np.hstack((vert_inds, vert_inds[np.zeros(extra, dtype='int')]))
        inds = np.hstack((horz_inds, vert_inds))
        assert inds.shape[0] % self.batch_size == 0, 'The number of samples here should be divisible by batch size'
        self.inds = inds
        self.size = len(self.inds)
