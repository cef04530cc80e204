"""MobileNetV2 Faster-RCNN for SNIPER, end to end (BASELINE config C1).

Same network, parameter names and graph outputs as the reference's symbols/faster/mobilenetv2_e2e.py (class contract
:152-163, trunk :184-225, RPN :164-173, heads :226-315 train / :316-377 test, initialiser :381-391), built table-driven
over the mx.sym API of sniper_amd.mx:

  data -> first 3x3/2 conv (32) + BN + relu6 -> [fp16]
       -> 7 inverted-residual sequences  (t, c, n, s) = (1,16,1,1) (6,24,2,2) (6,32,3,2) (6,64,4,2) (6,96,3,1) (6,160,3,2)
          (6,320,1,1); a block = 1x1 expand + BN + relu6 -> depthwise 3x3 + BN + relu6 -> 1x1 linear + BN (+ shortcut)
       -> last 1x1 conv (1280) + BN + relu6 -> [fp32]                                           stride 32: 512 -> 16
  RPN 3x3 (256) + relu -> cls(2A) / bbox(4A);   conv_new_1 1x1 (256) + relu -> D-PSROIPool x2 (1/32) -> fc512 x2 -> cls / bbox
"""
import numpy as np

import sniper_amd.mx as mx

from ..symbol import Symbol
from .resnet_mx_101_e2e import checkpoint_callback  # noqa: F401  (same epoch-end hook, reference :138-149)

# (expansion t, output channels c, repeats n, first stride s), reference :120-134
BOTTLENECKS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def relu6(data, prefix):
    return mx.sym.clip(data, 0, 6, name='%s-relu6' % prefix)


def mobilenet_unit(data, num_filter, kernel=1, stride=1, pad=0, num_group=1, if_act=True, prefix=''):
    conv = mx.sym.Convolution(data=data, num_filter=num_filter, kernel=(kernel, kernel), num_group=num_group,
                              stride=(stride, stride), pad=(pad, pad), no_bias=True, name='%s-conv2d' % prefix)
    bn = mx.sym.BatchNorm(data=conv, name='%s-batchnorm' % prefix, fix_gamma=False, momentum=0.995, eps=1e-5)
    return relu6(bn, prefix) if if_act else bn


def inverted_residual_unit(data, num_in_filter, num_filter, ifshortcut, stride, expansion_factor, prefix):
    nexp = int(round(num_in_filter * expansion_factor))
    x = mobilenet_unit(data, nexp, prefix='%s-exp' % prefix)
    x = mobilenet_unit(x, nexp, kernel=3, stride=stride, pad=1, num_group=nexp, prefix='%s-depthwise' % prefix)
    x = mobilenet_unit(x, num_filter, if_act=False, prefix='%s-linear' % prefix)
    if ifshortcut:
        return mx.sym.elemwise_add(data, x, name='%s-shortcut' % prefix)
    return x


class mobilenetv2_e2e(Symbol):
    def __init__(self, n_proposals=400, momentum=0.95, fix_bn=False, test_nbatch=1):
        Symbol.__init__(self)
        self.multiplier = 1
        self.test_nbatch = test_nbatch

    def get_bbox_param_names(self):
        return ['bbox_pred_weight', 'bbox_pred_bias']

    def get_rpn(self, conv_feat, num_anchors):
        conv = mx.sym.Convolution(data=conv_feat, kernel=(3, 3), pad=(1, 1), num_filter=256, name='rpn_conv_3x3')
        relu = mx.sym.Activation(data=conv, act_type='relu', name='rpn_relu')
        cls = mx.sym.Convolution(data=relu, kernel=(1, 1), pad=(0, 0), num_filter=2 * num_anchors, name='rpn_cls_score')
        box = mx.sym.Convolution(data=relu, kernel=(1, 1), pad=(0, 0), num_filter=4 * num_anchors, name='rpn_bbox_pred')
        return cls, box

    def _trunk(self, data):
        first_c = int(round(32 * self.multiplier))
        x = mobilenet_unit(data, first_c, kernel=3, stride=2, pad=1, prefix='first-3x3-conv')
        x = mx.sym.Cast(data=x, dtype=np.float16)
        in_c = first_c
        for i, (t, c, n, s) in enumerate(BOTTLENECKS):
            c = int(round(c * self.multiplier))
            x = inverted_residual_unit(x, in_c, c, False, s, t, 'seq-%d-block0' % i)
            for j in range(1, n):
                x = inverted_residual_unit(x, c, c, True, 1, t, 'seq-%d-block%d' % (i, j))
            in_c = c
        x = mobilenet_unit(x, int(1280 * self.multiplier) if self.multiplier > 1.0 else 1280, prefix='last-1x1-conv')
        return mx.sym.Cast(data=x, dtype=np.float32)

    def _head(self, feat, rois):
        """conv_new_1 -> two deformable PS-RoI poolings (the second with learned offsets) -> 2 FC -> cls / bbox."""
        conv_new_1 = mx.sym.Convolution(data=feat, kernel=(1, 1), num_filter=256, name='conv_new_1')
        relu = mx.sym.Activation(data=conv_new_1, act_type='relu', name='conv_new_1_relu')
        offset_t = mx.contrib.sym.DeformablePSROIPooling(name='offset_t', data=relu, rois=rois, group_size=1, pooled_size=7,
                                                         sample_per_part=4, no_trans=True, part_size=7, output_dim=256,
                                                         spatial_scale=0.03125)
        offset = mx.sym.FullyConnected(name='offset', data=offset_t, num_hidden=7 * 7 * 2, lr_mult=0.01)
        offset_reshape = mx.sym.Reshape(data=offset, shape=(-1, 2, 7, 7), name='offset_reshape')
        pool = mx.contrib.sym.DeformablePSROIPooling(name='deformable_roi_pool', data=relu, rois=rois, trans=offset_reshape,
                                                     group_size=1, pooled_size=7, sample_per_part=4, no_trans=False,
                                                     part_size=7, output_dim=256, spatial_scale=0.03125, trans_std=0.1)
        fc1 = mx.sym.Activation(data=mx.sym.FullyConnected(name='fc_new_1', data=pool, num_hidden=512), act_type='relu',
                                name='fc_new_1_relu')
        fc2 = mx.sym.Activation(data=mx.sym.FullyConnected(name='fc_new_2', data=fc1, num_hidden=512), act_type='relu',
                                name='fc_new_2_relu')
        return fc2

    def get_symbol_rcnn(self, cfg, is_train=True):
        num_anchors = cfg.network.NUM_ANCHORS
        num_classes = cfg.dataset.NUM_CLASSES
        data = mx.sym.Variable(name='data')
        im_info = mx.sym.Variable(name='im_info')
        if is_train:
            rpn_label = mx.sym.Variable(name='label')
            rpn_bbox_target = mx.sym.Variable(name='bbox_target')
            rpn_bbox_weight = mx.sym.Variable(name='bbox_weight')
            gt_boxes = mx.sym.Variable(name='gt_boxes')
            valid_ranges = mx.sym.Variable(name='valid_ranges')
            crowd_boxes = mx.sym.Variable(name='crowd_boxes')
        else:
            im_ids = mx.sym.Variable(name='im_ids')
            chip_ids = mx.sym.Variable(name='chip_ids')
        last_fm = self._trunk(data)
        rpn_cls_score, rpn_bbox_pred = self.get_rpn(last_fm, num_anchors)
        rpn_cls_score_reshape = mx.sym.Reshape(data=rpn_cls_score, shape=(0, 2, -1, 0), name='rpn_cls_score_reshape')
        if is_train:
            grad_scale = float(cfg.TRAIN.scale) if cfg.TRAIN.fp16 else 1.0
            B = cfg.TRAIN.BATCH_IMAGES
            rpn_cls_prob = mx.sym.SoftmaxOutput(data=rpn_cls_score_reshape, label=rpn_label, multi_output=True,
                                                normalization='valid', use_ignore=True, ignore_label=-1, name='rpn_cls_prob',
                                                grad_scale=grad_scale)
            rois, label, bbox_target, bbox_weight = mx.sym.MultiProposalTarget(
                cls_prob=rpn_cls_prob, bbox_pred=rpn_bbox_pred, im_info=im_info, gt_boxes=gt_boxes, valid_ranges=valid_ranges,
                crowd_boxes=crowd_boxes, batch_size=B, feature_stride=cfg.network.RPN_FEAT_STRIDE,
                scales=cfg.network.ANCHOR_SCALES, name='multi_proposal_target')
            label = mx.symbol.Reshape(data=label, shape=(-1,), name='label_reshape')
            fc2 = self._head(last_fm, rois)
            cls_score = mx.sym.FullyConnected(name='cls_score', data=fc2, num_hidden=num_classes)
            bbox_pred = mx.sym.FullyConnected(name='bbox_pred', data=fc2, num_hidden=4)
            cls_prob = mx.sym.SoftmaxOutput(name='cls_prob', data=cls_score, label=label, use_ignore=True, ignore_label=-1,
                                            grad_scale=grad_scale / (300.0 * B))
            bbox_loss_ = bbox_weight * mx.sym.smooth_l1(name='bbox_loss_', scalar=1.0, data=(bbox_pred - bbox_target))
            bbox_loss = mx.sym.MakeLoss(name='bbox_loss', data=bbox_loss_, grad_scale=grad_scale / (188.0 * B))
            cls_prob = mx.sym.Reshape(data=cls_prob, shape=(B, -1, num_classes), name='cls_prob_reshape')
            bbox_loss = mx.sym.Reshape(data=bbox_loss, shape=(B, -1, 4), name='bbox_loss_reshape')
            rpn_bbox_loss_ = rpn_bbox_weight * mx.sym.smooth_l1(name='rpn_bbox_loss_', scalar=1.0,
                                                                data=(rpn_bbox_pred - rpn_bbox_target))
            rpn_bbox_loss = mx.sym.MakeLoss(name='rpn_bbox_loss', data=rpn_bbox_loss_,
                                            grad_scale=3 * grad_scale / float(B * cfg.TRAIN.RPN_BATCH_SIZE))
            group = mx.sym.Group([rpn_cls_prob, rpn_bbox_loss, cls_prob, bbox_loss, mx.sym.BlockGrad(label)])
        else:
            rpn_cls_prob = mx.sym.SoftmaxActivation(data=rpn_cls_score_reshape, mode='channel', name='rpn_cls_prob')
            rpn_cls_prob_reshape = mx.sym.Reshape(data=rpn_cls_prob, shape=(0, 2 * num_anchors, -1, 0),
                                                  name='rpn_cls_prob_reshape')
            rois, _ = mx.sym.MultiProposal(cls_prob=rpn_cls_prob_reshape, bbox_pred=rpn_bbox_pred, im_info=im_info, name='rois',
                                           batch_size=self.test_nbatch, rpn_pre_nms_top_n=cfg.TEST.RPN_PRE_NMS_TOP_N,
                                           rpn_post_nms_top_n=cfg.TEST.RPN_POST_NMS_TOP_N, rpn_min_size=cfg.TEST.RPN_MIN_SIZE,
                                           threshold=cfg.TEST.RPN_NMS_THRESH, feature_stride=cfg.network.RPN_FEAT_STRIDE,
                                           ratios=tuple(cfg.network.ANCHOR_RATIOS), scales=tuple(cfg.network.ANCHOR_SCALES))
            fc2 = self._head(last_fm, rois)
            cls_score = mx.sym.FullyConnected(name='cls_score', data=fc2, num_hidden=num_classes)
            bbox_pred = mx.sym.FullyConnected(name='bbox_pred', data=fc2, num_hidden=4)
            cls_prob = mx.sym.SoftmaxActivation(name='cls_prob', data=cls_score)
            cls_prob = mx.sym.Reshape(data=cls_prob, shape=(self.test_nbatch, -1, num_classes), name='cls_prob_reshape')
            bbox_pred = mx.sym.Reshape(data=bbox_pred, shape=(self.test_nbatch, -1, 4), name='bbox_pred_reshape')
            group = mx.sym.Group([rois, cls_prob, bbox_pred, im_ids, im_info, chip_ids])
        self.sym = group
        return group

    def init_weight_rcnn(self, cfg, arg_params, aux_params):
        """N(0, 0.01) heads, zero biases, zero offset branch (reference :381-391)."""
        for name in ('rpn_conv_3x3', 'rpn_cls_score', 'rpn_bbox_pred', 'conv_new_1', 'fc_new_1', 'fc_new_2', 'cls_score',
                     'bbox_pred'):
            arg_params[name + '_weight'] = mx.random.normal(0, 0.01, shape=self.arg_shape_dict[name + '_weight'])
            arg_params[name + '_bias'] = mx.nd.zeros(shape=self.arg_shape_dict[name + '_bias'])
        arg_params['offset_weight'] = mx.nd.zeros(shape=self.arg_shape_dict['offset_weight'])
        arg_params['offset_bias'] = mx.nd.zeros(shape=self.arg_shape_dict['offset_bias'])
