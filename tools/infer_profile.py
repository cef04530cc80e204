"""Where the AutoFocus inference pass of bench.py (BASELINE C5) spends its time: cProfile of the second pass (executors bound and
cached), cumulative time per function.  GPU time shows up at the first synchronising call after the launches (asnumpy).

    python tools/infer_profile.py [n_top] [lanes] [batch sizes per scale, e.g. 8,8,8 (default: the yml's 8,8,2) or -] [concurrent jobs] [images per pass: the 8 synthetic images repeated] [freeze: gc.freeze() after pass 2 | -] [distinct: one array per image]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import sniper_amd.mx as mx
    from sniper_amd import config as cfgmod
    from sniper_amd.inference import imdb_detection_wrapper
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn
    from bench import focus_map_blobs

    class Imdb(object):
        num_classes, classes, name, result_path = 81, None, 'synthetic', None
    rs = np.random.RandomState(0)
    base = [{'image': rs.randint(0, 256, (480, 640, 3)).astype(np.uint8), 'width': 640, 'height': 480, 'flipped': False,
             'gt_overlaps': np.zeros((1, 81), np.float32)} for _ in range(8)]
    cfg = cfgmod.res101_e2e_autofocus()
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if len(sys.argv) > 3 and sys.argv[3] != '-':
        cfg.TEST.BATCH_IMAGES = tuple(int(b) for b in sys.argv[3].split(','))
    jobs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    n_img = int(sys.argv[5]) if len(sys.argv) > 5 else len(base)
    if len(sys.argv) > 7 and sys.argv[7] == 'distinct':          # every image its own array (its own upload), as bench.py does
        base = [dict(base[i % len(base)], image=rs.randint(0, 256, (480, 640, 3)).astype(np.uint8)) for i in range(n_img)]
    else:
        base = [base[i % len(base)] for i in range(n_img)]
    cache, blobs = {}, {}

    def fmap(scale_i, image, chip, net_map):        # (drawn once per (scale, image, chip), as bench.py does)
        key = (scale_i, image, chip, tuple(net_map.shape))
        if key not in blobs:
            blobs[key] = focus_map_blobs(scale_i, image, chip, net_map)
        return blobs[key]
    import gc
    freeze = len(sys.argv) > 6 and sys.argv[6] == 'freeze'
    gc_t = [0.0, 0.0, 0]

    def on_gc(phase, info):                      # time spent in the cyclic collector, per pass
        if phase == 'start':
            gc_t[1] = time.perf_counter()
        else:
            gc_t[0] += time.perf_counter() - gc_t[1]
            gc_t[2] += info['generation'] == 2
    gc.callbacks.append(on_gc)
    for p in range(9 if jobs > 1 else 7):                      # bind, capture, four replays timed plainly, one under cProfile
        roidb = [dict(r) for r in base]
        torch.cuda.synchronize()
        pr = cProfile.Profile() if p == (8 if jobs > 1 else 6) else None
        t0 = time.perf_counter()
        if pr:
            pr.enable()
        imdb_detection_wrapper(rn.resnet_mx_101_e2e, cfg, Imdb(), roidb, [mx.gpu(0)], None, None, module_cache=cache,
                               focus_map_fn=fmap, lanes=lanes, concurrent_jobs=jobs)
        torch.cuda.synchronize()
        if pr:
            pr.disable()
        print('pass %d: %.1f ms (cyclic collector: %.1f ms, %d full collections)' % (p, (time.perf_counter() - t0) * 1e3, gc_t[0] * 1e3,
                                                                                  gc_t[2]), flush=True)
        gc_t[0], gc_t[2] = 0.0, 0
        if freeze and p == 2:
            gc.collect()
            gc.freeze()
    mods = [m for k, m in cache.items() if hasattr(m, '_exes')] + [m for k, v in cache.items() if isinstance(k, tuple) and k and
                                                                   k[0] == '__lanes__' for m in v]
    print('bound executors per Module: %s; HBM held by this process: %.1f GB' % (
        [len(m._exes) for m in mods], torch.cuda.memory_allocated() / 1e9), flush=True)
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)


if __name__ == '__main__':
    main()
