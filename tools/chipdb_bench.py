"""Epoch chip database through the UNCHANGED reference path: `MNIteratorE2E.reset` (lib/iterators/MNIteratorE2E.py:40-103, the
lib2to3 artefact oracle/_ref/py3) with the reference's own `chip_worker` (lib/data_utils/data_workers.py:374-594) over the
sniper_amd extension mirrors, its `Pool(cfg.TRAIN.NUM_PROCESS)` being the drop-in pool of sniper_amd/ext/pool.py.

Timed twice on the SURVEY 8(d) synthetic roidb: with the pool's routing (each `pool.map(chip_worker.chip_extractor / box_assigner,
part)` = one ragged GPU launch) and, on a subset, with the routing off (`SNIPER_POOL_ROUTE=0`: every work item on a pool thread,
one kernel launch + read-back per image, the reference's Python loops under the interpreter lock -- the round-3 behaviour).  The
two must produce the same chip database from the same numpy seed.  Prints one JSON object (bench.py embeds it beside
`cpu_baseline`, whose Pool(64) of the reference's CPU workers builds the same database on the host cores).

    python tools/chipdb_bench.py [n_images (5000)] [n_images unrouted (400)]
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def build(n_images, route, seed=7):
    import numpy as np
    import torch
    from sniper_amd.synthetic import make_roidb
    import iterators.MNIteratorE2E as ref_it           # the reference's module (PY3/lib on sys.path), Pool patched by ext.install()
    import data_utils.data_workers as dw
    from configs.faster.default_configs import config, update_config
    update_config(os.path.join('configs', 'faster', 'sniper_res101_e2e.yml'))
    os.environ['SNIPER_POOL_ROUTE'] = '1' if route else '0'
    roidb = make_roidb(n_images, seed=0, n_proposals=0)
    for r in roidb:
        r['flipped'] = False
    config.TRAIN.USE_NEG_CHIPS = False                  # (negative-chip mining needs proposal files; SURVEY 8(d) C2 trains without)
    it = ref_it.MNIteratorE2E.__new__(ref_it.MNIteratorE2E)
    it.roidb, it.cfg, it.batch_size, it.epiter = roidb, config, 20, 0
    np.random.seed(seed)
    # un-routed: ONE pool thread, so that the work items draw their candidate permutations from numpy's global generator in
    # roidb order like the batched call does (with 64 threads -- or the reference's 64 forked processes, each with its own copy
    # of the generator -- the order, hence the database, differs from run to run)
    it.pool = ref_it.Pool(config.TRAIN.NUM_PROCESS if route else 1)
    it.chip_worker = dw.chip_worker(chip_size=512, cfg=config)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref_it.MNIteratorE2E.reset(it)                      # unchanged reference code
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it.pool.close()
    n_chips = int(sum(len(r['crops']) for r in roidb))
    return dt, n_chips, roidb, type(it.pool).__module__ + '.' + type(it.pool).__name__, getattr(it.pool, 'routed_maps', None)


def main():
    n_routed = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    n_plain = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    import numpy as np
    from acceptance_main_train import PY3, _install_environment
    work = tempfile.mkdtemp(prefix='sniper_chipdb_')
    _install_environment(work)
    os.chdir(PY3)
    for p in (PY3, os.path.join(PY3, 'lib')):
        sys.path.insert(0, p)
    import contextlib
    import io
    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet):
        build(64, True)                                 # warm-up: library load, first launches
        dt, chips, db, pool_name, routed = build(n_routed, True)
        dt_s, chips_s, db_s, _, routed_s = build(n_plain, True)
        dt_p, chips_p, db_p, _, routed_p = build(n_plain, False)
    same = chips_s == chips_p
    for a, b in zip(db_s, db_p):
        same = same and len(a['crops']) == len(b['crops']) and all(
            np.array_equal(np.asarray(x[0]), np.asarray(y[0])) and list(x[1:]) == list(y[1:]) for x, y in zip(a['crops'], b['crops']))
        same = same and len(a['props_in_chips']) == len(b['props_in_chips']) and all(
            np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a['props_in_chips'], b['props_in_chips']))
    out = {'value': round(chips / dt, 1), 'unit': 'chips/s', 'images': n_routed, 'chips': chips, 'seconds': round(dt, 3),
           'pool': pool_name, 'routed_maps': routed,
           'what': 'the reference\'s unchanged MNIteratorE2E.reset (chip extraction + box assignment of the epoch, '
                   'lib/iterators/MNIteratorE2E.py:40-103) over sniper_amd.ext: each pool.map of a chip_worker method is one ragged '
                   'GPU launch',
           'unrouted': {'value': round(chips_p / dt_p, 1), 'images': n_plain, 'seconds': round(dt_p, 3), 'routed_maps': routed_p,
                        'routed_same_subset_chips_per_s': round(chips_s / dt_s, 1),
                        'what': 'same call with SNIPER_POOL_ROUTE=0 on a pool of one thread: every work item its own kernel launch + read-back (the '
                                'round-3 behaviour; its 64 threads ran one at a time under the interpreter lock anyway)'},
           'routed_equals_unrouted': bool(same)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
