"""The reference's unchanged `MNIteratorE2E._get_batch` (lib/iterators/MNIteratorE2E.py:112-220; the lib2to3 artefact under
oracle/_ref/py3) over the drop-in pool, twice from the same iterator state: ROUTED (`pool.map(anchor_worker.worker, ...)` and
`thread_pool.map_async(im_worker.worker, ...)` as the mirrors' batched GPU work, the batch born in HBM through the shim's unplaced
`mx.nd.zeros`) and UNROUTED (SNIPER_POOL_ROUTE_BATCH=0: the reference's numpy workers per chip on pool threads, host arrays).
Prints one JSON object of comparisons (tests/test_gpu_acceptance.py asserts on it).  TEST INFRASTRUCTURE.

    python tools/routed_batch_check.py [n_images (24)] [batch (8)]
"""
import copy
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import numpy as np
    import torch
    from acceptance_main_train import PY3, _install_environment
    from fit_path_bench import _jpeg_roidb
    work = tempfile.mkdtemp(prefix='sniper_routed_')
    mx = _install_environment(work)
    os.chdir(PY3)
    for p in (PY3, os.path.join(PY3, 'lib')):
        sys.path.insert(0, p)
    import logging
    logging.getLogger('PIL').setLevel(logging.WARNING)
    import contextlib
    import io
    import iterators.MNIteratorE2E as ref_it
    from configs.faster.default_configs import config, update_config
    update_config(os.path.join('configs', 'faster', 'sniper_res101_e2e.yml'))
    config.TRAIN.USE_NEG_CHIPS = False
    # (one pool thread: the unrouted map then draws from numpy's global generator chip by chip IN ORDER, like a serial run of the reference)
    config.TRAIN.NUM_PROCESS, config.TRAIN.NUM_THREAD, config.TRAIN.CHIPS_DB_PARTS = 1, 4, 1
    roidb = _jpeg_roidb(work, n_images)
    for r in roidb:
        r['flipped'] = False
    np.random.seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        it = ref_it.MNIteratorE2E(roidb=roidb, config=config, batch_size=B, nGPUs=1, threads=4, pad_rois_to=400)
    out = {'pool': type(it.pool).__module__ + '.' + type(it.pool).__name__,
           'thread_pool': type(it.thread_pool).__module__ + '.' + type(it.thread_pool).__name__, 'batches': []}
    from sniper_amd.mx.ndarray import NDArray

    def host(a):
        return np.asarray(a.asnumpy() if isinstance(a, NDArray) else a, np.float32)

    def on_device(a):
        return isinstance(a, NDArray) and isinstance(a._store, torch.Tensor) and a._store.is_cuda
    for k in range(3):
        state = (it.cur_i, copy.deepcopy(it.crop_idx))
        routed0 = (it.pool.routed_maps, it.thread_pool.routed_maps)
        os.environ['SNIPER_POOL_ROUTE_BATCH'] = '1'
        br = it._get_batch()
        routed1 = (it.pool.routed_maps, it.thread_pool.routed_maps)
        it.cur_i, it.crop_idx = state[0], copy.deepcopy(state[1])
        os.environ['SNIPER_POOL_ROUTE_BATCH'] = '0'
        np.random.seed(100 + k)
        bu = it._get_batch()
        os.environ['SNIPER_POOL_ROUTE_BATCH'] = '1'
        # third call: routed again, with numpy's own sub-sampling draws replayed (SNIPER_NUMPY_RNG=1) under the seed of the unrouted call
        it.cur_i, it.crop_idx = state[0], copy.deepcopy(state[1])
        os.environ['SNIPER_NUMPY_RNG'] = '1'
        np.random.seed(100 + k)
        bn = it._get_batch()
        os.environ['SNIPER_NUMPY_RNG'] = '0'
        it.cur_i = state[0] + B
        n_lab, n_tgt, n_w, n_gt = [host(a) for a in bn.label[:4]]
        r_lab, r_tgt, r_w, r_gt = [host(a) for a in br.label[:4]]
        u_lab, u_tgt, u_w, u_gt = [host(a) for a in bu.label[:4]]
        rec = {'routed_maps': [routed1[0] - routed0[0], routed1[1] - routed0[1]],
               'routed_on_device': [bool(on_device(a)) for a in [br.data[0]] + list(br.label[:4])],
               'unrouted_on_device': [bool(on_device(a)) for a in [bu.data[0]] + list(bu.label[:4])],
               'gt_equal': bool(np.array_equal(r_gt, u_gt)),
               'valid_ranges_equal': bool(np.array_equal(host(br.data[1]), host(bu.data[1]))),
               'im_info_equal': bool(np.array_equal(host(br.data[2]), host(bu.data[2])))}
        # RPN labels: both paths keep all foreground anchors of a chip with fewer than num_fg of them (no draw involved) and fill
        # up to RPN_BATCH_SIZE with background drawn at random (numpy's generator / the kernel's hash: different draws)
        per_chip = []
        for i in range(B):
            fr, fu = r_lab[i] == 1, u_lab[i] == 1
            undrawn = int(fu.sum()) < 128 and int(fr.sum()) < 128
            per_chip.append({'fg': [int(fr.sum()), int(fu.sum())], 'bg': [int((r_lab[i] == 0).sum()), int((u_lab[i] == 0).sum())],
                             'fg_equal': bool(np.array_equal(fr, fu)) if undrawn else None,
                             'weights_equal': bool(np.array_equal(r_w[i], u_w[i])) if undrawn else None,
                             'targets_maxdiff': float(np.abs(r_tgt[i] - u_tgt[i]).max()) if undrawn else None})
        rec['chips'] = per_chip
        # with numpy's draws replayed every chip is comparable, drawn or not: labels (incl. the sub-sampled -1s), weights, targets
        ub_lab, ub_tgt, ub_w = [host(a) for a in bu.label[:3]]
        rec['numpy_rng'] = {'on_device': [bool(on_device(a)) for a in list(bn.label[:4])],
                            'labels_equal': bool(np.array_equal(n_lab, ub_lab)), 'weights_equal': bool(np.array_equal(n_w, ub_w)),
                            'gt_equal': bool(np.array_equal(n_gt, u_gt)),
                            'targets_maxdiff': float(np.abs(n_tgt - ub_tgt).max()),
                            'chips_with_a_draw': int(sum(1 for i in range(B) if (ub_lab[i] == -1).any() and
                                                         ((ub_lab[i] == 1).sum() + (ub_lab[i] == 0).sum()) == 256)),
                            'differs_from_hashed_draws': bool(not np.array_equal(n_lab, r_lab))}
        # pixels: the routed path resizes with OpenCV's fixed-point arithmetic, the unrouted one with the harness's PIL stand-in for
        # cv2 -- compared loosely (same crop, same scale, same mean subtraction: a shifted or mis-scaled chip would be far off)
        dr, du = host(br.data[0]), host(bu.data[0])
        rec['pixels_mean_abs_diff'] = float(np.abs(dr - du).mean())
        rec['pixels_corr'] = float(np.corrcoef(dr.reshape(-1)[::97], du.reshape(-1)[::97])[0, 1])
        out['batches'].append(rec)
    print(json.dumps(out), flush=True)
    os._exit(0)


if __name__ == '__main__':
    main()
