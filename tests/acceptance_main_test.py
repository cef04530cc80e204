"""Acceptance run (TEST INFRASTRUCTURE; needs a GPU and oracle/_ref/py3): the reference's OWN ``main_test.py`` executed over
sniper_amd -- ``main()`` (main_test.py:32-61) runs unchanged: ``update_config`` on ``configs/faster/sniper_res101_e2e.yml``
(three test scales, CONCURRENT_JOBS 2, soft-NMS), ``load_param(process=True)``, ``imdb_detection_wrapper`` ->
``detect_scale_worker`` (lib/inference.py:411-436: the reference's ``MNIteratorTestAutoFocus``, ``im_worker.worker_autofocus``,
``resnet_mx_101_e2e.get_symbol_rcnn(is_train=False)``, ``mx.mod.Module.bind / init_params / forward / get_outputs``),
``Tester.detect / get_detections / aggregate`` with ``nms_worker`` (-> the ``cpu_nms`` extension mirror, GPU soft-NMS).

Supplied by the harness, on top of what tests/acceptance_main_train.py supplies (mxnet / extension modules / cv2 / easydict /
dataset stand-ins):
  * ``load_proposal_roidb(..., get_imdb=True)`` returns a synthetic roidb (PNG files) and an imdb stub carrying the five
    attributes lib/inference.py reads (``num_classes, classes, name, result_path, evaluate_detections``);
  * a random-init checkpoint ``SNIPER-0007.params`` written through the shim where main_test.py looks for it, with the
    ``bbox_pred_{weight,bias}_test`` keys ``load_param(process=True)`` renames (train_utils/utils.py:96-99);
  * ``multiprocessing.Pool`` -> a serial in-process pool: the reference forks CONCURRENT_JOBS model processes and 32 NMS
    processes (lib/inference.py:459,159); forking after the HIP context exists is not possible, and the point here is the
    call sequence, not the process layout.

Checks written to the JSON: every scale produced detections for every image / class, the per-scale and final pickles
exist, the final detections respect MAX_PER_IMAGE and the image bounds, and a sample of the recorded
``nms_worker.worker`` calls equals the reference's compiled ``cpu_nms.pyx`` (oracle/_ref) on the same inputs.

    python tests/acceptance_main_test.py out.json
"""
import json
import os
import pickle
import runpy
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from acceptance_main_train import PY3, _install_environment, _synthetic_roidb  # noqa: E402


class _SerialPool(object):
    def __init__(self, *a, **k):
        pass

    def map(self, fn, items):
        return [fn(i) for i in items]

    def close(self):
        pass

    def join(self):
        pass


class _Imdb(object):
    """what lib/inference.py reads from lib/dataset/imdb.py objects (Tester.__init__ :58-62, imdb_detection_wrapper :464,527)"""
    def __init__(self, result_path):
        self.name = 'synthetic_val'
        self.num_classes = 81
        self.classes = ['__background__'] + ['c%02d' % i for i in range(1, 81)]
        self.result_path = result_path
        self.evaluated = None

    def evaluate_detections(self, all_boxes):
        self.evaluated = all_boxes
        return 'synthetic: no annotations to score'


def _checkpoint(mx, out_dir, epoch):
    """random-init full detector checkpoint in the layout the reference's checkpoint_callback writes (arg:/aux: keys,
    bbox_pred_*_test = the de-normalised regression weights, symbols/faster/resnet_mx_101_e2e.py checkpoint_callback)"""
    import numpy as np
    from sniper_amd import config as cfgmod
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as ours
    cfg = cfgmod.res101_e2e(batch_images=2)
    net = ours.resnet_mx_101_e2e(n_proposals=400, momentum=0.995)
    net.get_symbol_rcnn(cfg)
    net.infer_shape(dict(data=(2, 3, 512, 512), valid_ranges=(2, 2), im_info=(2, 3), label=(2, 21 * 32 * 32),
                         bbox_target=(2, 84, 32, 32), bbox_weight=(2, 84, 32, 32), gt_boxes=(2, 100, 5)))
    rs = np.random.RandomState(17)
    arg, aux = {}, {}
    new_layers = ('rpn_', 'conv_new_1', 'fc_new', 'cls_score', 'bbox_pred', 'offset', 'stage4_unit1_offset', 'stage4_unit2_offset',
                  'stage4_unit3_offset')
    for k, shp in net.arg_shape_dict.items():
        if k in ('data', 'valid_ranges', 'im_info', 'label', 'bbox_target', 'bbox_weight', 'gt_boxes'):
            continue
        if k.endswith('_gamma'):
            # a trained ResNet keeps its residual stream bounded; with unit gammas and identity running statistics 33 random
            # residual units double the variance each and leave the fp16 range: damp every unit's last BatchNorm instead
            v = np.full(shp, 0.3 if k.endswith('_bn3_gamma') else 1.0, np.float32)
        elif k.endswith(('_beta', '_bias')) or 'offset' in k:
            v = np.zeros(shp, np.float32)
        elif k.startswith(new_layers):
            # the detector heads as init_weight_rcnn leaves them (normal 0.01; resnet_mx_101_e2e.py:391-409), class scores a
            # little wider so that the 81 classes do not tie
            v = (rs.standard_normal(shp) * (0.05 if k.startswith('cls_score') else 0.01)).astype(np.float32)
        else:
            fan = int(np.prod(shp[1:]))
            v = (rs.standard_normal(shp) * np.sqrt(2.0 / fan)).astype(np.float32)
        arg[k] = mx.nd.array(v)
    for k, shp in net.aux_shape_dict.items():
        # bn_data normalises the mean-subtracted pixels (std ~ 74) so that the He-initialised trunk sees unit-variance input
        var = 5500.0 if k == 'bn_data_moving_var' else 1.0
        aux[k] = mx.nd.array(np.full(shp, var, np.float32) if k.endswith('_var') else np.zeros(shp, np.float32))
    assert 'bn_data_moving_var' in aux
    arg['bbox_pred_weight_test'] = mx.nd.array(arg['bbox_pred_weight'].asnumpy() * 0.1)
    arg['bbox_pred_bias_test'] = mx.nd.array(arg['bbox_pred_bias'].asnumpy() * 0.1)
    os.makedirs(out_dir)
    mx.model.save_checkpoint(os.path.join(out_dir, 'SNIPER'), epoch, None, arg, aux)
    return arg


def main(out_json, n_images=4, proposals=False):
    import numpy as np
    out_json = os.path.abspath(out_json)
    work = tempfile.mkdtemp(prefix='sniper_accept_test_')
    mx = _install_environment(work)
    import logging
    logging.getLogger('PIL').setLevel(logging.WARNING)       # create_logger puts the root logger at DEBUG (train_utils/utils.py:139-140)
    import multiprocessing
    multiprocessing.Pool = _SerialPool
    os.chdir(PY3)
    for p in (PY3, os.path.join(PY3, 'lib')):
        sys.path.insert(0, p)
    roidb = _synthetic_roidb(work, n_images)
    for r in roidb:                        # test roidbs are loaded with only_gt=True, flip=False (main_test.py:44-48)
        r['flipped'] = False
    output = os.path.join(work, 'output')
    imdb = _Imdb(os.path.join(output, 'results'))
    import data_utils.load_data as ld
    ld.load_proposal_roidb = lambda *a, **k: ([dict(r) for r in roidb], imdb)
    ckpt = _checkpoint(mx, os.path.join(output, 'sniper_res101_e2e', 'synthetic'), 7)

    # record the nms_worker calls of Tester.aggregate (inputs + outputs) to replay them on the reference's compiled cpu_nms
    import data_utils.data_workers as dw
    recorded = []
    ref_worker = dw.nms_worker.worker

    def recording_worker(self, data):
        keep = np.array(data, copy=True)
        out = ref_worker(self, data)
        if len(keep) > 1 and len(recorded) < 400:
            recorded.append((keep, np.array(out, copy=True)))
        return out

    dw.nms_worker.worker = recording_worker

    forwards, finite = [], []
    _fwd = mx.mod.Module.forward

    def forward(self, batch, is_train=None):
        forwards.append(tuple(int(v) for v in batch.data[0].shape))
        r = _fwd(self, batch, is_train=is_train)
        if proposals:
            return r
        outs = dict(zip(self.output_names, self.get_outputs()))
        cp = outs['cls_prob_reshape_output'].asnumpy()
        finite.append(bool(np.isfinite(cp).all() and np.isfinite(outs['bbox_pred_reshape_output'].asnumpy()).all()))
        if os.environ.get('ACCEPT_DEBUG'):
            print('DBG', forwards[-1], 'im_ids', outs['im_ids'].asnumpy(), 'chip', outs['chip_ids'].asnumpy(), 'im_info', batch.data[1].asnumpy().tolist(),
                  'cls finite', np.isfinite(cp).all(), 'per-image n>1e-3', [(cp[i][:, 1:] > 1e-3).sum() for i in range(cp.shape[0])],
                  'rois', outs['rois_output'].asnumpy()[::150].tolist(), flush=True)
        return r

    mx.mod.Module.forward = forward
    sys.argv = ['main_test.py', '--cfg', 'configs/faster/sniper_res101_e2e.yml', '--set', 'gpus', "'0'", 'output_path', output,
                'dataset.image_set', 'synthetic', 'dataset.test_image_set', 'synthetic_val']
    if proposals:
        # TEST.EXTRACT_PROPOSALS: imdb_proposal_extraction_wrapper / proposal_scale_worker / Tester.extract_proposals with the
        # reference's MNIteratorTest and get_symbol_rpn(is_train=False) (lib/inference.py:531-609,372-408)
        save = os.path.join(work, 'proposals')
        sys.argv += ['TEST.EXTRACT_PROPOSALS', 'True', 'TEST.PROPOSAL_SAVE_PATH', save]
    runpy.run_path(os.path.join(PY3, 'main_test.py'), run_name='__main__')
    mx.mod.Module.forward = _fwd
    if proposals:
        res = {'n_images': n_images, 'forward_shapes': sorted(set(forwards)), 'forwards': len(forwards)}
        files = os.listdir(save)
        res['proposal_files'] = files
        with open(os.path.join(save, files[0]), 'rb') as fh:
            props = pickle.load(fh)
        res['images'] = len(props)
        res['shapes'] = [list(np.asarray(p).shape) for p in props]
        res['dtype'] = str(np.asarray(props[0]).dtype)
        ok, sorted_scores = True, True
        for r, p in zip(roidb, props):
            p = np.asarray(p)
            ok &= bool(np.isfinite(p).all() and (p[:, 0] >= -1e-3).all() and (p[:, 1] >= -1e-3).all()
                       and (p[:, 2] <= r['width'] + 1e-3).all() and (p[:, 3] <= r['height'] + 1e-3).all()
                       and (p[:, 2] >= p[:, 0]).all() and (p[:, 3] >= p[:, 1]).all())
            for k in range(0, len(p), 300):              # N_PROPOSAL_PER_SCALE blocks, each in MultiProposal's score order
                sorted_scores &= bool((np.diff(p[k:k + 300, 4]) <= 1e-6).all())
        res.update(in_bounds=bool(ok), scores_sorted=bool(sorted_scores),
                   scale_files=sorted(d for d in os.listdir(imdb.result_path) if d.startswith('props_scale_')))
        with open(out_json, 'w') as fh:
            json.dump(res, fh, indent=1)
        print(json.dumps(res))
        return

    res = {'n_images': n_images, 'forward_shapes': sorted(set(forwards)), 'forwards': len(forwards), 'outputs_finite': all(finite)}
    final = imdb.evaluated
    res['evaluated'] = final is not None
    res['classes'] = len(final)
    res['images'] = len(final[1])
    per_image = [int(sum(len(final[j][i]) for j in range(1, 81))) for i in range(n_images)]
    res['dets_per_image'] = per_image
    ok_shape, in_bounds, finite = True, True, True
    for i, r in enumerate(roidb):
        for j in range(1, 81):
            d = np.asarray(final[j][i])
            ok_shape &= d.ndim == 2 and d.shape[1] == 5
            if len(d):
                finite &= bool(np.isfinite(d).all())
                in_bounds &= bool((d[:, 0] >= -1e-3).all() and (d[:, 1] >= -1e-3).all() and (d[:, 2] <= r['width'] + 1e-3).all()
                                  and (d[:, 3] <= r['height'] + 1e-3).all())
    res.update(shape_ok=bool(ok_shape), in_bounds=bool(in_bounds), finite=bool(finite))
    # pickles the reference wrote: per-scale detections / maps and the aggregate
    written = []
    for d, _, files in os.walk(imdb.result_path):
        for f in files:
            written.append(os.path.relpath(os.path.join(d, f), imdb.result_path))
    res['result_files'] = sorted(written)
    with open(os.path.join(imdb.result_path, 'dets_final', 'detections.pkl'), 'rb') as fh:
        back = pickle.load(fh)
    res['final_pickle_equal'] = bool(all(np.array_equal(np.asarray(back[j][i]), np.asarray(final[j][i]))
                                         for j in range(1, 81) for i in range(n_images)))
    # per-scale detections before aggregation: scores above the 1e-3 threshold, one entry per (class, image, chip)
    scales = {}
    for f in written:
        if f.startswith('dets_scale_') and f.endswith('detections.pkl'):
            with open(os.path.join(imdb.result_path, f), 'rb') as fh:
                sd = pickle.load(fh)
            n = [int(sum(len(c) for c in sd[j][i])) for i in range(n_images) for j in range(1, 81)]
            smin = min((float(np.asarray(c)[:, 4].min()) for j in range(1, 81) for i in range(n_images) for c in sd[j][i] if len(c)),
                       default=None)
            scales[os.path.dirname(f)] = {'dets': int(sum(n)), 'min_score': smin}
    res['scales'] = scales
    # the recorded soft-NMS calls against the reference's compiled Cython module
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from make_nms_golden import load_ref_cpu_nms
    refnms = load_ref_cpu_nms()
    cfg_sigma, n_cmp, n_equal, worst = 0.55, 0, 0, 0.0
    order = sorted(range(len(recorded)), key=lambda k: -len(recorded[k][0]))[:40]
    for k in order:
        dets, ours = recorded[k]
        ref = np.array(dets, dtype=np.float32, copy=True)
        ref = np.asarray(refnms.cpu_soft_nms(ref, np.float32(cfg_sigma), np.float32(0.3), np.float32(0.001), np.uint8(2)))   # lib/nms/nms.py:7-12
        n_cmp += 1
        if ref.shape == ours.shape:
            d = float(np.abs(ref - ours).max()) if ref.size else 0.0
            worst = max(worst, d)
            n_equal += int(d == 0.0)
    res['nms_replayed'] = n_cmp
    res['nms_bit_equal'] = n_equal
    res['nms_worst_absdiff'] = worst
    res['nms_sizes'] = [int(len(recorded[k][0])) for k in order[:5]]
    with open(out_json, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/tmp/acceptance_main_test.json', proposals='--proposals' in sys.argv[2:])
