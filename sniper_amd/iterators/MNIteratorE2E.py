"""SNIPER end-to-end training iterator on the GPU data path.

Contract of the reference's ``MNIteratorE2E`` (lib/iterators/MNIteratorE2E.py:16-220) +
``MNIteratorBase`` (MNIteratorBase.py:6-111): same constructor, ``reset()`` builds the epoch chip
database, every batch is an ``mx.io.DataBatch`` with

    data  = [data (B,3,512,512), valid_ranges (B,2), im_info (B,3) = h, w, im_scale]
    label = [label (B, A*F*F), bbox_target (B,4A,F,F), bbox_weight (B,4A,F,F), gt_boxes (B,100,5)]

and ``provide_data / provide_label / provide_*_single / batch_size / __len__ / get_batch_size``.
What differs is where the work happens: chip extraction + box assignment are two ragged GPU launches
per phase for the whole roidb (chip_worker), and the per-batch anchor labelling of all B chips is one
``sn_anchor_assign`` call whose outputs stay in HBM -- there is no multiprocessing pool and no pickling
(the reference's Pool(64).map round trips, :51-63,173).

Image pixels: the reference decodes and resizes JPEGs with OpenCV (im_worker, data_workers.py:80-121).  The default
source is the GPU ``im_worker`` (sniper_amd/data/im_worker.py: ``roidb[i]['image']`` decoded once, flip + crop + bilinear
resize + mean subtraction written straight into the batch tensor by ``sn_im_prepare``); a roidb entry whose image cannot
be opened is an error, not noise.  The benchmark is defined on synthetic chips:
``im_source='synthetic'`` asks for the seeded N(0, 50^2) generator explicitly (bench.py, the smoke test, the GPU tests);
a callable ``im_source(roidb_entry, crop, flipped) -> (3,H,W) float32`` is accepted too.
"""
import zlib

import numpy as np
import torch

import sniper_amd.mx as mx

from .. import hip
from ..data.anchors import AnchorAssigner
from ..data.chip_worker import chip_worker
from ..data.mask_utils import encode_chip_masks


def synthetic_im_source(chip_hw):
    """Seeded N(0, 50^2) chips keyed on (image, crop origin, scale index, flip): a CRC of the key, not hash() -- str hashes
    are salted per process, and every rank / run must see the same pixels."""
    def src(r, crop, flipped):
        key = repr((r.get('image') if isinstance(r.get('image'), str) else None, float(crop[0][0]), float(crop[0][1]),
                    int(crop[4]), bool(flipped)))
        rs = np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)
        return (rs.standard_normal((3, chip_hw[0], chip_hw[1])) * 50.0).astype(np.float32)
    return src


class MNIteratorE2E(mx.io.DataIter):
    def __init__(self, roidb, config, batch_size=4, threads=8, nGPUs=1, pad_rois_to=400, crop_size=(512, 512),
                 im_source=None):
        super(MNIteratorE2E, self).__init__()
        assert batch_size % nGPUs == 0, 'batch_size should be divisible by number of GPUs'
        self.roidb, self.cfg = roidb, config
        self.batch_size, self.crop_size = batch_size, crop_size
        self.n_per_gpu = batch_size // nGPUs
        self.data_name = ['data'] if config.TRAIN.ONLY_PROPOSAL else ['data', 'valid_ranges', 'im_info']
        self.label_name = ['label', 'bbox_target', 'bbox_weight'] if config.TRAIN.ONLY_PROPOSAL else \
            ['label', 'bbox_target', 'bbox_weight', 'gt_boxes']
        if config.TRAIN.AUTO_FOCUS:
            self.label_name.append('scale_label')      # FocusPixel labels (reference :28-29)
        if config.TRAIN.WITH_MASK:
            self.label_name.append('gt_masks')         # encoded polygons (B, 100, 500) (reference :31-32)
        self.chip_worker = chip_worker(config, crop_size[0])
        self.anchors = AnchorAssigner(config, crop_size[0])
        if im_source is None:           # the reference's behaviour: read roidb[i]['image'] (im_worker, data_workers.py:80-121)
            from ..data.im_worker import im_worker
            self.im_worker, self.im_source = im_worker(config, crop_size=crop_size[0]), None
        else:
            self.im_worker = None
            self.im_source = synthetic_im_source(crop_size) if im_source == 'synthetic' else im_source
        # the reference decodes the images of a batch on `threads` pool threads (self.thread_pool.map_async(im_worker.worker), :147);
        # here the pool only decodes + uploads the batch's images that are not yet resident (DeviceImageCache) -- the crop / resize
        # / mean subtraction of a chip is a kernel launch
        self.threads, self._decode_pool = max(1, int(threads)), None
        self.epiter = 0
        self.seed = 0
        self.batch = None
        self.reset()
        self.get_batch()

    # ---- MNIteratorBase contract ---------------------------------------------------------------
    def get_batch_size(self):
        return self.batch_size

    def __len__(self):
        return len(self.inds)

    @property
    def provide_data(self):
        return [(k, tuple(v.shape)) for k, v in zip(self.data_name, self.data)]

    @property
    def provide_label(self):
        return [(k, tuple(v.shape)) for k, v in zip(self.label_name, self.label)]

    provide_data_single = provide_data
    provide_label_single = provide_label

    def iter_next(self):
        return self.get_batch()

    def next(self):
        if self.iter_next():
            return self.batch
        raise StopIteration

    __next__ = next

    def getindex(self):
        return self.cur_i // self.batch_size

    def getpad(self):
        return 0

    # ---- epoch chip database (reference :41-103) ------------------------------------------------
    def reset(self):
        self.cur_i = 0
        self.n_neg_per_im = 2
        self.crop_idx = [0] * len(self.roidb)
        self.chip_worker.reset()
        crops = self.chip_worker.extract_batch(self.roidb)
        for r, cs in zip(self.roidb, crops):
            r['crops'] = cs
        assigned = self.chip_worker.assign_batch(self.roidb)
        for ps, r in zip(assigned, self.roidb):
            r['props_in_chips'] = list(ps[0])
            if self.cfg.TRAIN.USE_NEG_CHIPS:
                r['neg_crops'], r['neg_props_in_chips'] = ps[1], ps[2]
        chipindex = []
        for i, r in enumerate(self.roidb):
            if self.cfg.TRAIN.USE_NEG_CHIPS:
                cs = r['neg_crops']
                if len(cs) > 0:
                    sel = np.arange(len(cs))
                    if len(cs) > self.n_neg_per_im:
                        sel = np.random.permutation(sel)[0:self.n_neg_per_im]
                    for ind in sel:
                        r['crops'].append(r['neg_crops'][ind])
                        r['props_in_chips'].append(r['neg_props_in_chips'][ind].astype(np.int32))
            chipindex.extend([i] * len(r['crops']))
        chipindex = np.array(chipindex)
        self.n_chips = len(chipindex)
        if chipindex.shape[0] % self.batch_size > 0:
            extra = self.batch_size - (chipindex.shape[0] % self.batch_size)
            chipindex = np.hstack((chipindex, chipindex[0:extra]))
        self.inds = np.array(np.random.permutation(chipindex), dtype=int)
        for r in self.roidb:
            r['chip_order'] = np.random.permutation(np.arange(len(r['crops'])))
        self.epiter += 1
        self.size = len(self.inds)

    # ---- batch assembly (reference :112-220) ----------------------------------------------------
    def get_batch(self):
        if self.cur_i >= self.size:
            return False
        self.batch = self._get_batch()
        self.cur_i += self.batch_size
        return True

    def _get_batch(self):
        lo, hi = self.cur_i, self.cur_i + self.batch_size
        ids = [self.inds[i] for i in range(lo, hi)]
        roidb = [self.roidb[i] for i in ids]
        cropids = [r['chip_order'][self.crop_idx[i] % len(r['chip_order'])] for i, r in zip(ids, roidb)]
        for i in ids:
            self.crop_idx[i] += 1
        n = len(roidb)
        srange = np.zeros((n, 2), np.float32)
        chipinfo = np.zeros((n, 3), np.float32)
        worker_data = []
        dev = hip.require_gpu()
        if self.im_worker is not None:      # chips are written in HBM by sn_im_prepare, one launch per chip
            self._decode_ahead([r['image'] for r in roidb], dev)
            ims = torch.zeros((n, 3, self.crop_size[0], self.crop_size[1]), dtype=torch.float32, device=dev)
        else:
            ims = np.zeros((n, 3, self.crop_size[0], self.crop_size[1]), np.float32)
        for k, (r, cid) in enumerate(zip(roidb, cropids)):
            crop = r['crops'][cid]
            cur_crop, im_scale, height, width, scalei = crop
            nids = r['props_in_chips'][cid]
            gtids = np.where(r['max_overlaps'] == 1)[0]
            vr = self.cfg.TRAIN.VALID_RANGES[scalei]
            srange[k, 0] = 0 if vr[0] < 0 else vr[0] * im_scale
            srange[k, 1] = self.crop_size[1] if vr[1] < 0 else vr[1] * im_scale
            chipinfo[k] = [height, width, im_scale]
            worker_data.append([[self.crop_size[0], self.crop_size[1], im_scale], cur_crop, im_scale, nids, gtids,
                                r['boxes'][gtids, :], r['boxes'], r['max_classes'][gtids].reshape(-1, 1)])
            if self.im_worker is not None:
                self.im_worker.worker([r['image'], crop, r.get('flipped', False)], ims[k])
            else:
                im = self.im_source(r, crop, r.get('flipped', False))
                h, w = min(im.shape[1], self.crop_size[0]), min(im.shape[2], self.crop_size[1])
                ims[k, :, :h, :w] = im[:, :h, :w]
        out = self.anchors.assign(worker_data, seed=self.seed)
        self.seed += 1
        d_ims = ims if isinstance(ims, torch.Tensor) else torch.from_numpy(ims).to(dev)
        self.data = [mx.nd.NDArray(d_ims)] if self.cfg.TRAIN.ONLY_PROPOSAL else \
            [mx.nd.NDArray(d_ims), mx.nd.NDArray(torch.from_numpy(srange).to(dev)),
             mx.nd.NDArray(torch.from_numpy(chipinfo).to(dev))]
        self.label = [mx.nd.NDArray(out['label']), mx.nd.NDArray(out['bbox_target']), mx.nd.NDArray(out['bbox_weight'])]
        if not self.cfg.TRAIN.ONLY_PROPOSAL:
            self.label.append(mx.nd.NDArray(out['gt_boxes']))
        if self.cfg.TRAIN.AUTO_FOCUS:
            self.label.append(mx.nd.NDArray(self.anchors.focus_mask(worker_data)))      # scale_label (B, F*F), :182-197,213-214
        if self.cfg.TRAIN.WITH_MASK:                   # reference :186,199-200,216-217 (anchor_worker.worker :231-257)
            enc = np.stack([encode_chip_masks(w[0], w[1], w[2], w[5], w[7], r['gt_masks'])
                            for w, r in zip(worker_data, roidb)])
            self.label.append(mx.nd.NDArray(torch.from_numpy(enc).to(dev)))
        batch = mx.io.DataBatch(data=self.data, label=self.label, pad=0, index=self.getindex(),
                                provide_data=self.provide_data, provide_label=self.provide_label)
        batch.worker_data = worker_data   # the anchor-labelling inputs (bench.py re-runs the labelling per step)
        return batch


    def _decode_ahead(self, images, dev):
        from ..data.im_worker import decode_ahead
        decode_ahead(self.im_worker, images, self.threads)


# data parallel, one process per GPU: rank r assembles chips [cur_i + r B, cur_i + (r + 1) B) of every global batch
from ..ext.rank_slice import patch_iterator_class as _patch  # noqa: E402
_patch(MNIteratorE2E)
