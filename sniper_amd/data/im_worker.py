"""`im_worker` (lib/data_utils/data_workers.py:40-121) on the GPU: the decoded BGR uint8 image is uploaded once and
flip + crop + bilinear resize + mean subtraction + channel reversal + zero padding are one kernel (`sn_im_prepare`),
writing straight into the batch tensor in HBM -- no per-image float32 host arrays, no 8-thread pool.

Image source: the reference reads `roidb[i]['image']` with cv2.imread.  OpenCV is not part of this image; an entry may
be a numpy (H,W,3) uint8 BGR array, a `.npy` path, or any file PIL can open (converted RGB -> BGR)."""
import ctypes

import numpy as np
import torch

from .. import hip


def load_bgr(image):
    if isinstance(image, np.ndarray):
        im = image
    elif isinstance(image, str) and image.endswith('.npy'):
        im = np.load(image)
    else:
        from PIL import Image
        im = np.asarray(Image.open(image).convert('RGB'))[:, :, ::-1]
    im = np.ascontiguousarray(im, np.uint8)
    if im.ndim != 3 or im.shape[2] != 3:
        raise ValueError('expected an (H, W, 3) BGR uint8 image, got %s' % (im.shape,))
    return im


def target_scale(width, height, target_size):
    """Scale rule shared by im_worker.worker (:99-105), MNIteratorTestAutoFocus (:46-51) and add_chips."""
    im_size_min, im_size_max = min(width, height), max(width, height)
    scale = float(target_size[0]) / float(im_size_min)
    if np.round(scale * im_size_max) > target_size[1]:
        scale = float(target_size[1]) / float(im_size_max)
    return scale


def _to_device(im):
    """uint8 (H, W, 3) array -> device tensor.  (A ring of pinned staging slots with asynchronous copies was measured against this
    plain copy on the 64-image pass: 390 - 420 ms instead of 286 - 300, profiles/r04_infer_long_pass.txt -- not adopted.)"""
    return hip.dev(im)


class DeviceImageCache(object):
    """Decoded uint8 images resident on the device, by path (or by object for in-memory arrays).  The reference reads and decodes
    an image once per test scale (lib/data_utils/data_workers.py:49-78 under every scale's iterator); a coarse-to-fine pass visits
    every image at three scales, so `imdb_detection_wrapper` hands ONE cache to the iterators of all its scales: one decode and one
    upload per image and pass (a 5000-image pass of 640 x 480 images is 4.6 GB of a 288 GB card).  Bounded in bytes
    (SNIPER_IMAGE_CACHE_GB, default 16): the oldest entries go first."""

    def __init__(self, max_bytes=None):
        import os
        import threading
        self.max_bytes = int(float(os.environ.get('SNIPER_IMAGE_CACHE_GB', '16')) * (1 << 30)) if max_bytes is None else int(max_bytes)
        self._d, self._bytes, self._lock = {}, 0, threading.Lock()
        self.hits = self.misses = 0

    def get(self, image):
        key = image if isinstance(image, str) else id(image)
        with self._lock:
            ent = self._d.get(key)
            if ent is not None and (isinstance(image, str) or ent[0] is image):
                self.hits += 1
                return ent[1]
        d = _to_device(load_bgr(image))
        n = d.numel() * d.element_size()
        with self._lock:
            ent = self._d.get(key)
            if ent is not None and (isinstance(image, str) or ent[0] is image):
                self.hits += 1                 # another pool thread uploaded the same image meanwhile (chips of one image share a
                return ent[1]                  # batch): keep its tensor, this copy is dropped
            old = self._d.pop(key, None)       # an id() reused by another array: the stale entry's bytes leave the count
            if old is not None:
                self._bytes -= old[2]
            self.misses += 1
            while self._d and self._bytes + n > self.max_bytes:
                gone = self._d.pop(next(iter(self._d)))
                self._bytes -= gone[2]
            # (the array itself is kept with an id() key: the id of a collected array can be reused by another image)
            self._d[key] = (None if isinstance(image, str) else image, d, n)
            self._bytes += n
        return d

    def holds(self, image):
        key = image if isinstance(image, str) else id(image)
        with self._lock:
            ent = self._d.get(key)
            return ent is not None and (isinstance(image, str) or ent[0] is image)

    def __len__(self):
        return len(self._d)


_DECODE_POOLS = {}


def decode_ahead(worker, images, threads=8):
    """Decode + upload the distinct images of a batch that `worker`'s device cache does not hold yet, on `threads` threads (JPEG /
    PNG decoding releases the interpreter lock; one image is 3-10 ms, a batch of 20 chips meets ~3-20 new ones).  The threads bind
    the caller's device and CURRENT stream (the prefetch worker's own), so the uploads do not wait behind the training step."""
    cache = getattr(worker, '_cache', None)
    need, seen = [], set()
    for im in images:
        key = im if isinstance(im, str) else id(im)
        if key in seen or (cache is not None and cache.holds(im)):
            continue
        seen.add(key)
        need.append(im)
    if len(need) < 2 or threads <= 1:
        return
    # ONE pool per (device, thread count) for the process -- not one per caller stream: every PrefetchingIter worker has a stream of
    # its own, and a pool keyed on it leaked eight threads per iterator ever built.  The stream is bound per TASK.
    dev, st = torch.cuda.current_device(), torch.cuda.current_stream()
    key = (dev, int(threads))
    pool = _DECODE_POOLS.get(key)
    if pool is None:
        from multiprocessing.pool import ThreadPool

        def init():
            torch.cuda.set_device(dev)
        pool = _DECODE_POOLS[key] = ThreadPool(int(threads), initializer=init)

    def task(im):
        with torch.cuda.stream(st):
            return worker._device_image(im)
    pool.map(task, need, chunksize=1)


class im_worker(object):
    def __init__(self, cfg, crop_size=None, target_size=None, image_cache=None):
        self.cfg = cfg
        self.crop_size = crop_size
        self.target_size = target_size if target_size else cfg.TRAIN.SCALES[0]
        self.means = np.asarray(cfg.network.PIXEL_MEANS, np.float64)      # (uint8 - float64 in the reference: the subtraction is done in double)
        self._cache = image_cache if image_cache is not None else DeviceImageCache(256 * 4 << 20)

    def _device_image(self, image):
        return self._cache.get(image)

    def _run(self, image, out, crop, scale, flip):
        d = self._device_image(image)
        H, W = int(d.shape[0]), int(d.shape[1])
        hw = (ctypes.c_int32 * 2)()
        x1, y1, x2, y2 = crop if crop is not None else (0, 0, W, H)
        hip.call('sn_im_prepare', d, H, W, int(x1), int(y1), int(x2), int(y2), float(scale), 1 if flip else 0,
                 self.means.ctypes.data_as(ctypes.c_void_p), out, int(out.shape[1]), int(out.shape[2]), hw, hip.stream())
        return int(hw[0]), int(hw[1])

    def worker_autofocus(self, data, out):
        """data = [image, max_size, flipped, crop (x1,y1,x2,y2) or None, scale]; out (3, Hm, Wm) device view.
        -> (scale, (resized_h, resized_w)) like the reference (:49-78) (the tensor is written in place)."""
        image, max_size, flipped, crop, scale = data
        c = None if crop is None else (max(int(crop[0]), 0), max(int(crop[1]), 0), int(crop[2]), int(crop[3]))
        return scale, self._run(image, out, c, scale, False)

    def worker(self, data, out):
        """data = [image, crop-or-max_size, flipped]; full-image mode computes the scale from target_size (:99-105)."""
        image, second, flipped = data[0], data[1], data[2]
        if self.crop_size:
            crop = second
            c = (int(crop[0][0]), int(crop[0][1]), int(crop[0][2]), int(crop[0][3]))
            self._run(image, out, c, crop[1], flipped)
            return None
        d = self._device_image(image)
        scale = target_scale(int(d.shape[1]), int(d.shape[0]), self.target_size)
        return scale, self._run(image, out, None, scale, flipped)
