"""Acceptance run (TEST INFRASTRUCTURE; needs a GPU and oracle/_ref/py3): the reference's OWN ``main_train.py`` executed
over sniper_amd -- its ``__main__`` block (main_train.py:36-146) runs unchanged: ``update_config`` on the reference's
``configs/faster/sniper_res101_e2e.yml``, the reference's ``MNIteratorE2E`` / ``PrefetchingIter`` (lib/iterators), its
``chip_worker`` / ``anchor_worker`` / ``im_worker`` (lib/data_utils/data_workers.py), ``symbols/faster/resnet_mx_101_e2e.py``,
``lib/train_utils/{metric,utils,lr_scheduler}.py``, ``mx.mod.Module.fit`` with the reference's callbacks and checkpointing.

What the harness supplies (and nothing else):
  * ``mxnet``           -> sniper_amd.mx (the shim under test), ``chips`` / ``bbox`` / ``cpu_nms`` / ``gpu_nms`` -> sniper_amd.ext
    (the extension-module mirrors under test; ``bbox`` and ``chips`` additionally keep the reference's pure-Python
    sub-modules importable, as the reference's in-package .so files do);
  * ``cv2``             -> a PIL-backed stand-in for imread / resize (OpenCV is not installed; image pixels are not compared);
  * ``easydict``        -> the stand-in of sniper_amd.config; ``yaml.load`` gets the Loader argument PyYAML >= 6 demands;
  * ``dataset``         -> empty module, and ``load_proposal_roidb`` returns a synthetic COCO-shaped roidb whose images
    are PNG files written to a scratch directory (there is no COCO here; lib/dataset is out of scope);
  * a random-init "pretrained" checkpoint written through the shim (there is no ImageNet file here);
  * nothing for ``multiprocessing.Pool``: ``sniper_amd.ext.install()`` itself makes ``from multiprocessing import Pool`` hand
    the reference a thread-backed pool (sniper_amd/ext/pool.py) -- its 64 forked worker PROCESSES would each have called the
    GPU-backed extension modules from a copy of this process's HIP context.

Checks (written to the JSON the pytest wrapper asserts on): the epoch ran to the end over the reference iterator, the
reference's metrics are finite, parameters moved, the reference's checkpoint callbacks wrote loadable files, and the
labels of the recorded ``anchor_worker.worker`` calls (numpy RNG seeded per call) equal sniper_amd's GPU anchor labelling
of the same chips bit for bit (labels, weights; targets to 1e-6).

    python tests/acceptance_main_train.py out.json
"""
import copy
import json
import os
import runpy
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY3 = os.path.join(ROOT, 'oracle', '_ref', 'py3')


def _install_environment(work):
    import numpy as np
    for name, val in (('float', float), ('int', int)):
        if not hasattr(np, name):
            setattr(np, name, val)
    import yaml
    _load = yaml.load
    yaml.load = lambda stream, Loader=None: _load(stream, Loader=Loader or yaml.FullLoader)

    sys.path.insert(0, ROOT)
    import sniper_amd.mx as mx
    mx.alias_as('mxnet')
    from sniper_amd import config as cfgmod
    ed = types.ModuleType('easydict')
    ed.EasyDict = cfgmod.AttrDict
    sys.modules['easydict'] = ed
    sys.modules['dataset'] = types.ModuleType('dataset')

    from sniper_amd import ext
    ext.install()
    from sniper_amd.ext import bbox as ext_bbox, chips as ext_chips
    for name, mod, fns in (('bbox', ext_bbox, ('bbox_overlaps_cython', 'ignore_overlaps_cython')), ('chips', ext_chips, ('generate',))):
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(PY3, 'lib', name)]
        for f in fns:
            setattr(pkg, f, getattr(mod, f))
        sys.modules[name] = pkg

    from PIL import Image
    cv2 = types.ModuleType('cv2')
    cv2.IMREAD_COLOR, cv2.INTER_LINEAR, cv2.IMREAD_IGNORE_ORIENTATION = 1, 1, 128

    def imread(path, flags=1):
        return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1])      # BGR like OpenCV

    def resize(im, dsize=None, dst=None, fx=None, fy=None, interpolation=1):
        h, w = im.shape[:2]
        nw, nh = (int(round(w * fx)), int(round(h * fy))) if not dsize else dsize
        return np.asarray(Image.fromarray(np.ascontiguousarray(im.astype(np.uint8))).resize((max(nw, 1), max(nh, 1)), Image.BILINEAR))

    cv2.imread, cv2.resize = imread, resize
    sys.modules['cv2'] = cv2
    return mx


def _synthetic_roidb(work, n_images):
    import numpy as np
    from PIL import Image
    from sniper_amd.synthetic import make_roidb
    roidb = make_roidb(n_images, seed=3, n_proposals=60)
    rs = np.random.RandomState(5)
    os.makedirs(os.path.join(work, 'images'))
    for i, r in enumerate(roidb):
        path = os.path.join(work, 'images', '%06d.png' % i)
        Image.fromarray(rs.randint(0, 256, (r['height'], r['width'], 3)).astype(np.uint8)).save(path)
        r['image'] = path
        k = len(r['boxes'])
        ov = np.zeros((k, 81), np.float32)
        ov[np.arange(k), r['max_classes']] = r['max_overlaps']
        r['gt_overlaps'] = ov
    return roidb


def _pretrained(mx, work):
    """random-init backbone checkpoint 'resnet-0000.params' in the reference's format (arg: / aux: keys)"""
    import numpy as np
    from sniper_amd import config as cfgmod
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as ours
    cfg = cfgmod.res101_e2e(batch_images=2)
    net = ours.resnet_mx_101_e2e(n_proposals=400, momentum=0.995)
    sym = net.get_symbol_rcnn(cfg)
    net.infer_shape(dict(data=(2, 3, 512, 512), valid_ranges=(2, 2), im_info=(2, 3), label=(2, 21 * 32 * 32),
                         bbox_target=(2, 84, 32, 32), bbox_weight=(2, 84, 32, 32), gt_boxes=(2, 100, 5)))
    rs = np.random.RandomState(11)
    new_layers = ('rpn_', 'conv_new_1', 'fc_new', 'cls_score', 'bbox_pred', 'offset', 'stage4_unit1_offset', 'stage4_unit2_offset', 'stage4_unit3_offset')
    arg, aux = {}, {}
    for k, shp in net.arg_shape_dict.items():
        if k in ('data', 'valid_ranges', 'im_info', 'label', 'bbox_target', 'bbox_weight', 'gt_boxes') or k.startswith(new_layers):
            continue
        if k.endswith('_gamma'):
            v = np.ones(shp, np.float32)
        elif k.endswith(('_beta', '_bias')):
            v = np.zeros(shp, np.float32)
        else:
            fan = int(np.prod(shp[1:]))
            v = (rs.standard_normal(shp) * np.sqrt(2.0 / fan)).astype(np.float32)
        arg[k] = mx.nd.array(v)
    for k, shp in net.aux_shape_dict.items():
        aux[k] = mx.nd.array(np.ones(shp, np.float32) if k.endswith('_var') else np.zeros(shp, np.float32))
    prefix = os.path.join(work, 'pretrained', 'resnet')
    os.makedirs(os.path.dirname(prefix))
    mx.model.save_checkpoint(prefix, 0, None, arg, aux)
    return prefix


def main(out_json, n_images=3):
    import numpy as np
    out_json = os.path.abspath(out_json)
    work = tempfile.mkdtemp(prefix='sniper_accept_')
    mx = _install_environment(work)
    os.chdir(PY3)
    for p in (PY3, os.path.join(PY3, 'lib')):
        sys.path.insert(0, p)
    roidb = _synthetic_roidb(work, n_images)
    import data_utils.load_data as ld
    ld.load_proposal_roidb = lambda *a, **k: [dict(r) for r in roidb]
    prefix = _pretrained(mx, work)

    # record the reference anchor_worker's inputs / outputs (numpy RNG seeded per call so the labelling can be replayed)
    import data_utils.data_workers as dw
    recorded = []
    ref_worker = dw.anchor_worker.worker

    def recording_worker(self, data):
        seed = 1000 + len(recorded)
        np.random.seed(seed)
        keep = copy.deepcopy(data) if len(recorded) < 6 else None      # the worker shifts / scales its box arrays in place
        out = ref_worker(self, data)
        if keep is not None:
            recorded.append((seed, keep, out))
        return out

    dw.anchor_worker.worker = recording_worker

    batches = []
    sys.argv = ['main_train.py', '--cfg', 'configs/faster/sniper_res101_e2e.yml', '--display', '1', '--set',
                'gpus', "'0'", 'output_path', os.path.join(work, 'output'), 'network.pretrained', prefix,
                'dataset.image_set', 'synthetic', 'TRAIN.BATCH_IMAGES', '2', 'TRAIN.end_epoch', '1', 'TRAIN.NUM_PROCESS', '1',
                'TRAIN.NUM_THREAD', '1', 'TRAIN.CHIPS_DB_PARTS', '1']
    # count the batches Module.fit consumes (the reference's Speedometer is its batch_end_callback; we only observe)
    _fit = mx.mod.Module.fit

    def fit(self, train_data, *a, **k):
        cb = k.get('batch_end_callback')
        cbs = cb if isinstance(cb, (list, tuple)) else [cb]

        def count(param):      # ahead of the reference's Speedometer, which resets the metrics after it has logged them
            names, vals = param.eval_metric.get()
            batches.append((int(param.nbatch), {n: float(v) for n, v in zip(names, vals)}))
        k['batch_end_callback'] = [count] + [c for c in cbs if c is not None]
        return _fit(self, train_data, *a, **k)

    mx.mod.Module.fit = fit
    g = runpy.run_path(os.path.join(PY3, 'main_train.py'), run_name='__main__')
    mx.mod.Module.fit = _fit

    res = {'batches': len(batches), 'iterator': type(g['train_iter']).__module__ + '.' + type(g['train_iter']).__name__,
           'symbol': type(g['sym_inst']).__module__, 'n_chips': int(len(g['train_iter'])) if hasattr(g['train_iter'], '__len__') else None}
    res['metrics'] = batches[-1][1]          # running values at the last batch of the epoch
    res['metrics_first'] = batches[0][1]
    # parameters moved away from the checkpoint they were initialised from
    arg_now, aux_now = g['mod'].get_params()
    arg0 = g['arg_params']
    moved = {}
    for k in ('rpn_conv_3x3_weight', 'stage3_unit1_conv1_weight', 'fc_new_1_weight', 'stage1_unit1_conv1_weight'):
        moved[k] = float(np.abs(arg_now[k].asnumpy() - arg0[k].asnumpy()).max())
    res['param_delta'] = moved
    # the reference's epoch-end callbacks wrote checkpoints
    outdir = g['output_path']
    files = sorted(os.listdir(outdir))
    res['checkpoint_files'] = files
    params = [f for f in files if f.endswith('.params')]
    if params:
        loaded = mx.nd.load(os.path.join(outdir, params[-1]))
        res['checkpoint_keys'] = len(loaded)
        res['checkpoint_has_test_weights'] = any(k.endswith('bbox_pred_weight_test') for k in loaded)
    # recorded anchor_worker calls vs the GPU labelling of sniper_amd (bit-exact with the numpy RNG replayed)
    from sniper_amd.data.anchors import AnchorAssigner
    cfg = g['config']
    aa = AnchorAssigner(cfg, 512)
    cmp = []
    for seed, data, out in recorded:
        pre = aa.assign([data], want_label_pre=True)
        lp = pre['label_pre'].cpu().numpy()
        np.random.seed(seed)
        keys = aa.numpy_replay_keys(lp[0:1])
        ours = aa.assign([data], keys=keys)
        _np = lambda v: v.asnumpy() if hasattr(v, 'asnumpy') else np.asarray(v)
        label = _np(out[0]).reshape(-1)
        # the reference worker returns sparse targets: values + index triplets (data_workers.py:354-362); densify them the way
        # MNIteratorE2E.py:189-194 does
        tv, pid = np.asarray(out[1], np.float32), _np(out[2]).astype(int)
        dense_t = np.zeros(tuple(ours['bbox_target'][0].shape), np.float32)
        dense_w = np.zeros_like(dense_t)
        if pid.size:
            dense_t[pid[0], pid[1], pid[2]] = tv
            dense_w[pid[0], pid[1], pid[2]] = 1.0
        cmp.append({
            'label_equal': bool(np.array_equal(ours['label'][0].cpu().numpy().reshape(-1).astype(np.float32), label.astype(np.float32))),
            'weight_equal': bool(np.array_equal(ours['bbox_weight'][0].cpu().numpy(), dense_w)),
            'target_maxdiff': float(np.abs(ours['bbox_target'][0].cpu().numpy() - dense_t).max()),
            'gt_equal': bool(np.array_equal(ours['gt_boxes'][0].cpu().numpy(), _np(out[3]).astype(np.float32))),
            'n_fg': int((label == 1).sum()), 'n_bg': int((label == 0).sum()),
            'label_ndiff': int((ours['label'][0].cpu().numpy().reshape(-1) != label.astype(np.float32)).sum()),
            'ours_fg_bg': [int((ours['label'][0] == 1).sum()), int((ours['label'][0] == 0).sum())],
            'gt_maxdiff': float(np.abs(ours['gt_boxes'][0].cpu().numpy() - _np(out[3]).astype(np.float32)).max()),
            'gt_rows': [int((_np(out[3])[:, 0] >= 0).sum()), int((ours['gt_boxes'][0].cpu().numpy()[:, 0] >= 0).sum())],
        })
    res['anchor_labels'] = cmp
    with open(out_json, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/tmp/acceptance_main_train.json')
