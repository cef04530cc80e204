"""CPU: the executor's lowering pass (formats, fusions, parameter arenas) without launching kernels."""
import numpy as np
import torch

import sniper_amd.mx as mx
from sniper_amd import config as cfgmod
from sniper_amd.engine.executor import Executor
from sniper_amd.symbols.faster import resnet_mx_101_e2e as ours
from sniper_amd.train import fixed_param_names


def test_r101_lowering_plan():
    B = 2
    cfg = cfgmod.res101_e2e(batch_images=B)
    net = ours.resnet_mx_101_e2e(momentum=0.995)
    sym = net.get_symbol_rcnn(cfg)
    shapes = dict(data=(B, 3, 512, 512), valid_ranges=(B, 2), im_info=(B, 3), label=(B, 21 * 32 * 32),
                  bbox_target=(B, 84, 32, 32), bbox_weight=(B, 84, 32, 32), gt_boxes=(B, 100, 5))
    ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym), device=torch.device('cpu'))
    kinds = {}
    for s in ex.steps:
        kinds[type(s).__name__] = kinds.get(type(s).__name__, 0) + 1
    assert kinds['ConvolutionStep'] == 33 * 3 + 4 + 1 + 3 + 1 + 3 - 3     # bottleneck convs + shortcuts + stem + rpn + conv_new_1 + offsets, minus the 3 deformable conv2
    assert kinds['DeformableConvolutionStep'] == 3 and kinds['DPSROIPoolStep'] == 2 and kinds['FullyConnectedStep'] == 5
    bns = [s for s in ex.steps if type(s).__name__ == 'BatchNormStep']
    assert sum(1 for s in bns if s.relu) == 100 and sum(1 for s in bns if s.is_stem) == 1
    # loss-side tensors are fp32, everything else fp16 channels-last
    f32 = sorted(s.node.name for s in ex.steps if type(s).__name__ in ('ConvolutionStep', 'FullyConnectedStep') and s.y.fmt == 'f32')
    assert f32 == ['bbox_pred', 'cls_score', 'offset', 'rpn_bbox_pred', 'rpn_cls_score']
    # frozen stem + stage1: no gradients requested there
    for s in ex.steps:
        if type(s).__name__ == 'ConvolutionStep' and (s.node.name == 'conv0' or s.node.name.startswith('stage1')):
            assert not s.y.needs_grad
        if type(s).__name__ == 'ConvolutionStep' and s.node.name.startswith('stage3'):
            assert s.y.needs_grad
    assert abs(ex.n_trainable / 1e6 - 73.48) < 0.05
    assert set(g[0][:2] for g in ex.groups) == {(0.01, 0.0), (0.01, 1.0), (1.0, 0.0), (1.0, 1.0)}   # offset fc lr_mult .01; bias/beta wd 0
    assert len(ex.groups) <= 8
    # the weights the reference holds in fp16 (stage2-4 convolutions incl. the offset convs) lead the arena: 43.3 M of
    # 73.5 M; their gradients cross xGMI in fp16
    assert abs(ex.half_elems / 1e6 - 42.6) < 1.0, ex.half_elems
    assert ex.params['stage3_unit5_conv2_weight'].half_region and ex.params['stage4_unit1_offset_weight'].half_region
    assert not ex.params['rpn_conv_3x3_weight'].half_region and not ex.params['fc_new_1_weight'].half_region
    assert not ex.params['stage3_unit5_bn2_gamma'].half_region
    assert all(p.offset < ex.half_elems for p in ex.params.values() if p.trainable and p.half_region)
    assert all(p.offset >= ex.half_elems for p in ex.params.values() if p.trainable and not p.half_region)
    assert ex.params['fc_new_1_weight'].int_shape == (1024, 49, 256)
    assert ex.params['rpn_conv_3x3_weight'].int_shape == (512, 9, 3072)
    # reference <-> kernel layout round trip
    p = ex.params['fc_new_1_weight']
    a = np.random.RandomState(0).standard_normal(p.ref_shape).astype(np.float32)
    assert np.array_equal(p.to_reference(p.to_internal(a)), a)
    p = ex.params['stage3_unit1_conv2_weight']
    a = np.random.RandomState(0).standard_normal(p.ref_shape).astype(np.float32)
    assert np.array_equal(p.to_reference(p.to_internal(a)), a)


def test_test_graph_lowers():
    cfg = cfgmod.res101_e2e(batch_images=2)
    net = ours.resnet_mx_101_e2e(test_nbatch=2)
    sym = net.get_symbol_rcnn(cfg, is_train=False)
    shapes = dict(data=(2, 3, 512, 512), im_info=(2, 3), im_ids=(2,), chip_ids=(2,))
    ex = Executor(sym, shapes, False, [], device=torch.device('cpu'))
    assert any(type(s).__name__ == 'MultiProposalStep' for s in ex.steps)
    assert ex.n_trainable == 0
    # test-time BatchNorm folding: bn2 / bn3 of every bottleneck and bn0 (the packed stem convolution) read a convolution nobody
    # else reads -> folded into it; bn1 reads the residual sum (also the next shortcut's input), bn_data the image: not folded
    bns = [s for s in ex.steps if type(s).__name__ == 'BatchNormStep']
    folded = sorted(s.node.name for s in bns if s.folded_into is not None)
    # (the three stage-4 bn3 layers read a DeformableConvolution: sampling + GEMM, no epilogue to fold into)
    assert len(folded) == 33 * 2 - 3 + 1 and all(n.endswith(('_bn2', '_bn3')) or n == 'bn0' for n in folded), folded[:5]
    assert not any(n.startswith('stage4') and n.endswith('_bn3') for n in folded)
    assert all(s.folded_into.fold_bn is s and s.y.t is s.x.t for s in bns if s.folded_into is not None)
    assert not any(type(s).__name__ == 'DeformableConvolutionStep' and getattr(s, 'fold_bn', None) is not None for s in ex.steps)


def test_batchnorm_folding_test_time_and_frozen_training_layers(monkeypatch):
    cfg = cfgmod.res101_e2e(batch_images=2)
    net = ours.resnet_mx_101_e2e(momentum=0.995)
    sym = net.get_symbol_rcnn(cfg)
    shapes = dict(data=(2, 3, 512, 512), valid_ranges=(2, 2), im_info=(2, 3), label=(2, 21 * 32 * 32),
                  bbox_target=(2, 84, 32, 32), bbox_weight=(2, 84, 32, 32), gt_boxes=(2, 100, 5))
    ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym), device=torch.device('cpu'))
    # a training graph folds ONLY frozen moving-statistics layers behind frozen convolutions (network.FIXED_PARAMS = conv0, bn0,
    # stage1): conv0 -> bn0 and conv1 -> bn2, conv2 -> bn3 of the three stage-1 units; every batch-statistics layer keeps its pass
    folded = sorted(s.node.name for s in ex.steps if getattr(s, 'folded_into', None) is not None)
    assert folded == sorted(['bn0'] + ['stage1_unit%d_bn%d' % (u, b) for u in (1, 2, 3) for b in (2, 3)]), folded
    for s in ex.steps:
        if getattr(s, 'folded_into', None) is not None:
            assert s.global_stats and not s.folded_into.w.trainable and not s.y.needs_grad
    monkeypatch.setenv('SNIPER_TRAIN_FOLD_BN', '0')
    ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym), device=torch.device('cpu'))
    assert not any(getattr(s, 'folded_into', None) is not None for s in ex.steps)
    monkeypatch.setenv('SNIPER_INFER_FOLD_BN', '0')
    net = ours.resnet_mx_101_e2e(test_nbatch=2)
    sym = net.get_symbol_rcnn(cfg, is_train=False)
    ex = Executor(sym, dict(data=(2, 3, 512, 512), im_info=(2, 3), im_ids=(2,), chip_ids=(2,)), False, [], device=torch.device('cpu'))
    assert not any(getattr(s, 'folded_into', None) is not None for s in ex.steps)


def test_mobilenetv2_lowering_plan():
    """BASELINE C1 graph: depthwise convolutions, BN+relu6 fusion, the trainable packed stem, stride-32 heads."""
    from sniper_amd.symbols.faster import mobilenetv2_e2e as mn
    B = 2
    cfg = cfgmod.mobilenetv2_e2e(batch_images=B)
    net = mn.mobilenetv2_e2e()
    sym = net.get_symbol_rcnn(cfg)
    shapes = dict(data=(B, 3, 512, 512), valid_ranges=(B, 2), im_info=(B, 3), label=(B, 15 * 16 * 16),
                  bbox_target=(B, 60, 16, 16), bbox_weight=(B, 60, 16, 16), gt_boxes=(B, 100, 5), crowd_boxes=(B, 10, 5))
    ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym), device=torch.device('cpu'))
    convs = [s for s in ex.steps if type(s).__name__ == 'ConvolutionStep']
    assert len(convs) == 57 and sum(1 for s in convs if s.depthwise) == 17 and sum(1 for s in convs if s.is_stem) == 1
    bns = [s for s in ex.steps if type(s).__name__ == 'BatchNormStep']
    assert len(bns) == 53 and sum(1 for s in bns if s.act == 2) == 36          # every BN but the 17 linear ones feeds a relu6
    assert all(s.fused for s in ex.steps if type(s).__name__ == 'ClipStep')
    # FIXED_PARAMS of the MobileNetV2 config freeze every gamma/beta and nothing else: all convolutions train
    assert all(s.w.trainable for s in convs)
    assert not any(p.trainable for n, p in ex.params.items() if n.endswith('_gamma') or n.endswith('_beta'))
    stem = ex.params['first-3x3-conv-conv2d_weight']
    assert stem.kind == 'stem' and stem.int_shape == (32, 3, 16)
    a = np.random.RandomState(0).standard_normal(stem.ref_shape).astype(np.float32)
    assert np.array_equal(stem.to_reference(stem.to_internal(a)), a)
    dw = ex.params['seq-3-block1-depthwise-conv2d_weight']
    assert dw.int_shape == (384, 9, 1) and dw.wT16 is None


def test_rfcn_lowering_plan():
    """BASELINE C4 graph: position-sensitive R-FCN head (group_size 7) on the R101 trunk."""
    from sniper_amd.symbols.faster import resnet_mx_101_e2e_rfcn as rf
    B = 2
    cfg = cfgmod.res101_e2e(batch_images=B)
    net = rf.resnet_mx_101_e2e_rfcn(momentum=0.995)
    sym = net.get_symbol_rcnn(cfg)
    shapes = dict(data=(B, 3, 512, 512), valid_ranges=(B, 2), im_info=(B, 3), label=(B, 21 * 32 * 32),
                  bbox_target=(B, 84, 32, 32), bbox_weight=(B, 84, 32, 32), gt_boxes=(B, 100, 5))
    ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym), device=torch.device('cpu'))
    ps = [s for s in ex.steps if type(s).__name__ == 'DPSROIPoolStep']
    assert len(ps) == 4 and all(s.G == 7 and s.P == 7 for s in ps)
    assert sorted(s.D for s in ps) == [2, 2, 4, 81]
    assert sum(1 for s in ps if s.trans is not None) == 2
    assert not any(type(s).__name__ == 'FullyConnectedStep' for s in ex.steps)
    votes = [s for s in ex.steps if type(s).__name__ == 'PoolingStep' and s.kind == 'gavg']
    assert len(votes) == 2 and all(s.y.fmt == 'f32' and s.y.needs_grad for s in votes)
    assert ex.params['rfcn_cls_weight'].ref_shape == (49 * 81, 256, 1, 1)
    assert ex.shapes[(id(sym._heads[-3][0]), 0)] == (B, 300, 81)          # cls_prob_reshape
    net.infer_shape(shapes)
    arg, aux = {}, {}
    net.init_weight_rcnn(cfg, arg, aux)
    assert float(np.abs(arg['rfcn_cls_offset_t_weight'].asnumpy()).max()) == 0.0
    assert arg['rfcn_bbox_weight'].shape == (49 * 4, 256, 1, 1)


def test_mask_lowering_plan():
    """sniper_res101_e2e_mask.yml graph: RPN on C4, mask RoIs, 14x14 pooling, 4 convs, 2x2 deconvolution as 1x1 conv + shuffle,
    per-RoI channel picks, per-pixel 2-way softmax."""
    from sniper_amd.symbols.faster import resnet_mx_101_e2e_mask as mk
    B = 2
    cfg = cfgmod.res101_e2e(batch_images=B)
    net = mk.resnet_mx_101_e2e_mask(momentum=0.995)
    sym = net.get_symbol_rcnn(cfg)
    shapes = dict(data=(B, 3, 512, 512), valid_ranges=(B, 2), im_info=(B, 3), label=(B, 21 * 32 * 32),
                  bbox_target=(B, 84, 32, 32), bbox_weight=(B, 84, 32, 32), gt_boxes=(B, 100, 5), gt_masks=(B, 100, 500))
    ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym), device=torch.device('cpu'))
    kinds = {}
    for s in ex.steps:
        kinds[type(s).__name__] = kinds.get(type(s).__name__, 0) + 1
    assert kinds['MultiProposalTargetMaskStep'] == 1 and kinds['MaskRcnnTargetStep'] == 1 and kinds['DeconvolutionStep'] == 1
    assert kinds['PickStep'] == 2 and kinds['DPSROIPoolStep'] == 4
    pool = [s for s in ex.steps if type(s).__name__ == 'DPSROIPoolStep' and s.P == 14]
    assert len(pool) == 2 and all(s.rois.shape == (B * 50, 5) for s in pool)
    picks = [s for s in ex.steps if type(s).__name__ == 'PickStep']
    assert all(s.y.fmt == 'act' and s.y.shape == (B * 50, 1, 28, 28) and s.idx.fmt == 'f32' for s in picks)
    dec = [s for s in ex.steps if type(s).__name__ == 'DeconvolutionStep'][0]
    assert dec.y.shape == (B * 50, 256, 28, 28) and dec.w.kind == 'deconv' and dec.w.int_shape == (1024, 1, 256) and dec.w.wT16 is not None
    a = np.random.RandomState(0).standard_normal(dec.w.ref_shape).astype(np.float32)
    assert dec.w.ref_shape == (256, 256, 2, 2) and np.array_equal(dec.w.to_reference(dec.w.to_internal(a)), a)
    i = dec.w.to_internal(a)                                  # row (a, b, o) holds W[:, o, a, b]
    assert np.array_equal(i[(1 * 2 + 0) * 256 + 7, 0], a[:, 7, 1, 0])
    assert ex.params['rpn_conv_3x3_weight'].int_shape == (512, 9, 1024)
    out = [s for s in ex.steps if type(s).__name__ == 'SoftmaxOutputStep' and s.node.name == 'mask_cls_prob'][0]
    assert ex.shapes[(id(out.node), 0)] == (B * 50, 2, 28, 28)


def test_backward_split_for_allreduce_overlap():
    """Data parallel: the backward pass is cut where about half of its GEMM work is done; the parameters behind the cut
    (heads, RPN, stage 4) lead the gradient arena and are all-reduced while the rest (stages 2-3) still runs."""
    B = 2
    cfg = cfgmod.res101_e2e(batch_images=B)
    sym = ours.resnet_mx_101_e2e(momentum=0.995).get_symbol_rcnn(cfg)
    shapes = dict(data=(B, 3, 512, 512), valid_ranges=(B, 2), im_info=(B, 3), label=(B, 21 * 32 * 32),
                  bbox_target=(B, 84, 32, 32), bbox_weight=(B, 84, 32, 32), gt_boxes=(B, 100, 5))
    ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym), device=torch.device('cpu'), split_backward=True)
    assert ex.split_k > 0
    names = [s.node.name for s in ex.steps]
    first_late = names[ex.split_k]
    assert first_late.startswith('stage4') or first_late.startswith('stage3_unit2'), first_late   # the stage-3 / stage-4 boundary region
    ph = {n: p.phase for n, p in ex.params.items() if p.trainable}
    assert ph['fc_new_1_weight'] == 0 and ph['rpn_conv_3x3_weight'] == 0 and ph['stage4_unit3_conv2_weight'] == 0
    assert ph['stage2_unit1_conv1_weight'] == 1 and ph['stage3_unit2_conv1_weight'] == 1 and ph['stage3_unit2_bn1_gamma'] == 1
    # arena: phase 0 first, inside a phase the fp16-transported weights first; ranges tile the arena
    assert [r[0] for r in ex.ar_ranges] == sorted(r[0] for r in ex.ar_ranges)
    assert ex.ar_ranges[0][2] == 0 and all(a[3] == b[2] for a, b in zip(ex.ar_ranges, ex.ar_ranges[1:])) and ex.ar_ranges[-1][3] == ex.n_trainable
    early = sum(b - a for p_, h, a, b in ex.ar_ranges if p_ == 0)
    assert 0.5 < early / ex.n_trainable < 0.9            # most parameter bytes are final with half of the backward to go
    for n, p in ex.params.items():
        if p.trainable:
            r = [q for q in ex.ar_ranges if q[2] <= p.offset < q[3]][0]
            assert (r[0], r[1]) == (p.phase, p.half_region), n
    assert len(ex.groups) <= 16
    # without the split nothing changes
    ex0 = Executor(sym, shapes, True, fixed_param_names(cfg, sym), device=torch.device('cpu'))
    assert ex0.split_k == 0 and all(p.phase == 0 for p in ex0.params.values()) and [r[:2] for r in ex0.ar_ranges] == [[0, True], [0, False]]
