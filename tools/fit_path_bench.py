"""What a user of the drop-in gets: the reference's OWN main_train.py (main_train.py:36-146, unchanged -- update_config on its
yml, its metrics (lib/train_utils/metric.py), callbacks, Speedometer, `PrefetchingIter` + `mod.fit`) at the BASELINE batch of
20 chips per GPU on a synthetic COCO-shaped roidb whose images are files on disk; chips/s from WALL time over the batches after a
warm-up (bind, two eager steps, hipGraph capture).  VERDICT r4 item 4.

    python tools/fit_path_bench.py [mirror|reference] [batch=20] [timed_batches=50] [--profile]

  mirror      `iterators.MNIteratorE2E` / `iterators.PrefetchingIter` resolve to sniper_amd/iterators -- the lib/iterators API surface
              this engine keeps (north_star), GPU data path: sn_im_prepare per chip from the device image cache, one sn_anchor_assign
              per batch, batches born in HBM.  This is the drop-in's number (`fit_path.value` on the bench line).
  reference   the reference's own lib/iterators + lib/data_utils workers (numpy anchor labelling, cv2 stand-in resize) on the drop-in
              pool's threads, batches assembled on the host and copied up by the Module: the CPU data path's rate, reported beside it.

The harness (tests/acceptance_main_train.py::_install_environment) supplies what the acceptance test supplies: `mxnet` ->
sniper_amd.mx, the extension-module mirrors, a PIL-backed `cv2`, `easydict`, an empty `dataset` module, the synthetic roidb, a
random-init "pretrained" file.  Prints ONE JSON line.  TEST / MEASUREMENT INFRASTRUCTURE: runs the lib2to3 translation of the
reference under oracle/_ref/py3 as the CALLER of the product; nothing here is imported by sniper_amd/."""
import json
import os
import runpy
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class _Enough(Exception):
    pass


def _jpeg_roidb(work, n_images):
    """COCO-shaped synthetic roidb whose images are JPEG files like COCO's: smooth noise (8 x 8 blocks, upsampled)
    so that a file is ~60-100 KB and decodes in the few ms a photograph does -- uniform per-pixel noise is the worst case of
    every codec."""
    import numpy as np
    from PIL import Image
    from sniper_amd.synthetic import make_roidb
    roidb = make_roidb(n_images, seed=3, n_proposals=60)
    rs = np.random.RandomState(5)
    os.makedirs(os.path.join(work, 'images'))
    for i, r in enumerate(roidb):
        path = os.path.join(work, 'images', '%06d.jpg' % i)
        h, w = r['height'], r['width']
        small = rs.randint(0, 256, ((h + 7) // 8, (w + 7) // 8, 3)).astype(np.uint8)
        Image.fromarray(small).resize((w, h), Image.BILINEAR).save(path, quality=90)
        r['image'] = path
        k = len(r['boxes'])
        ov = np.zeros((k, 81), np.float32)
        ov[np.arange(k), r['max_classes']] = r['max_overlaps']
        r['gt_overlaps'] = ov
    return roidb


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    mode = args[0] if args else 'mirror'
    batch = int(args[1]) if len(args) > 1 else 20
    timed = int(args[2]) if len(args) > 2 else 50
    profile = '--profile' in sys.argv
    warm = 8
    import numpy as np
    import torch
    import acceptance_main_train as acc
    PY3 = acc.PY3
    if not os.path.isdir(PY3):
        print(json.dumps({'value': None, 'sample': 'oracle/_ref/py3 (the translated reference mains) is not present'}))
        return
    work = tempfile.mkdtemp(prefix='sniper_fitpath_')
    mx = acc._install_environment(work)
    os.chdir(PY3)
    for p in (PY3, os.path.join(PY3, 'lib')):
        sys.path.insert(0, p)
    # ~6 chips per image at three scales: enough images for warm + timed batches in ONE epoch
    n_images = max(8, int(np.ceil((warm + timed + 4) * batch / 5.0)))
    roidb = _jpeg_roidb(work, n_images)
    import logging
    logging.getLogger('PIL').setLevel(logging.WARNING)      # (the reference's create_logger turns the ROOT logger to DEBUG)
    import data_utils.load_data as ld
    ld.load_proposal_roidb = lambda *a, **k: [dict(r) for r in roidb]
    prefix = acc._pretrained(mx, work)
    if mode == 'mirror':
        import importlib
        import iterators            # the reference's package (namespace for its other members)
        for name in ('MNIteratorE2E', 'PrefetchingIter'):
            m = importlib.import_module('sniper_amd.iterators.' + name)
            sys.modules['iterators.' + name] = m
            setattr(iterators, name, m)
    sys.argv = ['main_train.py', '--cfg', 'configs/faster/sniper_res101_e2e.yml', '--display', '10', '--set',
                'gpus', "'0'", 'output_path', os.path.join(work, 'output'), 'network.pretrained', prefix,
                'dataset.image_set', 'synthetic', 'TRAIN.BATCH_IMAGES', str(batch), 'TRAIN.end_epoch', '1', 'TRAIN.NUM_PROCESS', '8',
                'TRAIN.NUM_THREAD', '8', 'TRAIN.CHIPS_DB_PARTS', '1']
    stamps = []
    _fit = mx.mod.Module.fit

    def fit(self, train_data, *a, **k):
        cb = k.get('batch_end_callback')
        cbs = [c for c in (cb if isinstance(cb, (list, tuple)) else [cb]) if c is not None]

        def stamp(param):           # BEHIND the reference's Speedometer: the batch's host work is done
            stamps.append(time.perf_counter())
            if len(stamps) == warm:
                torch.cuda.synchronize()
                stamps[-1] = time.perf_counter()
            if len(stamps) >= warm + timed:
                torch.cuda.synchronize()
                stamps[-1] = time.perf_counter()
                raise _Enough()
        k['batch_end_callback'] = cbs + [stamp]
        fit.iterator = type(train_data).__module__ + '.' + type(train_data).__name__
        fit.inner = type(train_data.iters[0]).__module__ if hasattr(train_data, 'iters') else None
        return _fit(self, train_data, *a, **k)
    mx.mod.Module.fit = fit
    prof = None
    if profile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t_start = time.perf_counter()
    try:
        runpy.run_path(os.path.join(PY3, 'main_train.py'), run_name='__main__')
        ended = 'epoch end'
    except _Enough:
        ended = 'stopped after %d batches' % len(stamps)
    if prof is not None:
        prof.disable()
        import pstats
        import io
        s = io.StringIO()
        pstats.Stats(prof, stream=s).sort_stats('cumulative').print_stats(45)
        sys.stderr.write(s.getvalue())
    mx.mod.Module.fit = _fit
    n = len(stamps) - warm
    out = {'mode': mode, 'batch': batch, 'iterator': getattr(fit, 'iterator', None), 'inner_iterator_module': getattr(fit, 'inner', None),
           'batches_total': len(stamps), 'ended': ended, 'setup_and_warmup_s': round((stamps[warm - 1] if len(stamps) >= warm else time.perf_counter()) - t_start, 2)}
    if n > 0:
        dt = stamps[-1] - stamps[warm - 1]
        gaps = np.diff(np.asarray(stamps[warm - 1:])) * 1e3
        out.update(value=round(n * batch / dt, 2), unit='chips/s', batches=n, seconds=round(dt, 3), ms_per_batch=round(dt / n * 1e3, 3),
                   ms_per_batch_median=round(float(np.median(gaps)), 3), ms_per_batch_max=round(float(gaps.max()), 3),
                   what='reference main_train.py unchanged (PrefetchingIter + mod.fit + 6 EvalMetrics + Speedometer), BATCH_IMAGES %d, '
                        '%d images on disk, wall clock over %d batches after %d warm-up batches' % (batch, n_images, n, warm))
    else:
        out.update(value=None, sample='the epoch ended before the warm-up did (%d batches)' % len(stamps))
    print(json.dumps(out), flush=True)
    os._exit(0)            # (the reference's logger / prefetch thread: nothing to wait for)


if __name__ == '__main__':
    main()
