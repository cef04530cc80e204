"""ResNet-101 (pre-activation) C4 + deformable-dilated C5 Faster-RCNN for SNIPER, end to end.

Same network, parameter names and graph outputs as the reference's
symbols/faster/resnet_mx_101_e2e.py (class contract :20-34, trunk :394-448, heads :227-392), built
table-driven over the mx.sym API of sniper_amd.mx.  Structure:

  data -> bn_data -> conv0 7x7/2 -> [fp16] bn0 relu maxpool3x3/2
       -> stage1 (3 units, 256, frozen BN) -> stage2 (4 units, 512, /2) -> stage3 (23 units, 1024, /2)   = conv_feat
       -> stage4 (3 deformable-dilated units, 2048, stride 1)
  cat4 = concat(conv_feat, stage4) [fp32] -> RPN 3x3+relu -> cls(2A) / bbox(4A)
                                          -> conv_new_1 1x1 (256) -> D-PSROIPool x2 -> fc1024 x2 -> cls / bbox
"""
import numpy as np

import sniper_amd.mx as mx

from ..symbol import Symbol

BBOX_STDS = (0.1, 0.1, 0.2, 0.2)


def checkpoint_callback(bbox_param_names, prefix, means, stds):
    """Epoch-end hook (reference :6-17): additionally stores bbox_pred_{weight,bias}_test = params
    scaled by the fixed target stds so that inference decodes un-normalised deltas."""
    def _callback(iter_no, sym, arg, aux):
        wn, bn = bbox_param_names
        if wn not in arg:
            return
        s = np.array(BBOX_STDS)
        arg[wn + '_test'] = (arg[wn].T * mx.nd.array(s)).T
        arg[bn + '_test'] = arg[bn] * mx.nd.array(s)
        mx.model.save_checkpoint(prefix, iter_no + 1, sym, arg, aux)
        arg.pop(wn + '_test')
        arg.pop(bn + '_test')
    return _callback


class resnet_mx_101_e2e(Symbol):
    UNITS = (3, 4, 23, 3)
    WIDTHS = (64, 256, 512, 1024, 2048)

    def __init__(self, n_proposals=400, momentum=0.95, fix_bn=False, test_nbatch=1):
        Symbol.__init__(self)
        self.momentum, self.fix_bn, self.test_nbatch = momentum, fix_bn, test_nbatch
        self.units, self.filter_list = self.UNITS, list(self.WIDTHS)
        self.workspace = 512

    def get_bbox_param_names(self):
        return ['bbox_pred_weight', 'bbox_pred_bias']

    # ---- building blocks -----------------------------------------------------------------------
    def _bn(self, x, name, frozen):
        if frozen or self.fix_bn:
            return mx.sym.BatchNorm(data=x, name=name, fix_gamma=False, eps=2e-5, use_global_stats=True)
        return mx.sym.BatchNorm(data=x, name=name, fix_gamma=False, eps=2e-5, momentum=self.momentum)

    def _bn_relu(self, x, prefix, k, frozen):
        y = self._bn(x, '%s_bn%d' % (prefix, k), frozen)
        return mx.sym.Activation(data=y, act_type='relu', name='%s_relu%d' % (prefix, k))

    def _conv(self, x, name, nf, k, stride=1, pad=0, dilate=1, bias=False):
        return mx.sym.Convolution(data=x, name=name, num_filter=nf, kernel=(k, k), stride=(stride, stride), pad=(pad, pad),
                                  dilate=(dilate, dilate), no_bias=not bias, workspace=self.workspace)

    def _unit(self, x, nf, stride, dim_match, name, frozen=False, deform=False, dilate=False):
        """Pre-activation bottleneck (reference :36-145).  conv2 is a plain 3x3, a dilation-2 3x3
        (residual_unit_dilate) or a 4-group deformable 3x3 with a learned 72-channel offset field."""
        mid = int(nf * 0.25)
        a1 = self._bn_relu(x, name, 1, frozen)
        c1 = self._conv(a1, name + '_conv1', mid, 1)
        a2 = self._bn_relu(c1, name, 2, frozen)
        if deform:
            off = mx.sym.Convolution(data=a2, name=name + '_offset', num_filter=72, kernel=(3, 3), stride=(1, 1), pad=(2, 2),
                                     dilate=(2, 2), cudnn_off=True)
            c2 = mx.contrib.sym.DeformableConvolution(data=a2, offset=off, name=name + '_conv2', num_filter=512, kernel=(3, 3),
                                                      stride=(1, 1), pad=(2, 2), dilate=(2, 2), num_deformable_group=4,
                                                      no_bias=True)
        elif dilate:
            c2 = self._conv(a2, name + '_conv2', mid, 3, stride, 2, 2)
        else:
            c2 = self._conv(a2, name + '_conv2', mid, 3, stride, 1)
        a3 = self._bn_relu(c2, name, 3, frozen)
        c3 = self._conv(a3, name + '_conv3', nf, 1)
        sc = x if dim_match else self._conv(a1, name + '_sc', nf, 1, stride)
        return c3 + sc

    def resnetc4(self, data, fp16=False):
        x = mx.sym.BatchNorm(data=data, name='bn_data', fix_gamma=True, eps=2e-5, use_global_stats=True)
        x = self._conv(x, 'conv0', self.filter_list[0], 7, 2, 3)
        if fp16:
            x = mx.sym.Cast(data=x, dtype=np.float16)
        x = mx.sym.BatchNorm(data=x, name='bn0', fix_gamma=False, eps=2e-5, use_global_stats=True)
        x = mx.sym.Activation(data=x, act_type='relu', name='relu0')
        x = mx.sym.Pooling(data=x, kernel=(3, 3), stride=(2, 2), pad=(1, 1), pool_type='max')
        for s in range(3):
            nf, frozen = self.filter_list[s + 1], s == 0
            for u in range(self.units[s]):
                x = self._unit(x, nf, (1 if s == 0 else 2) if u == 0 else 1, u > 0, 'stage%d_unit%d' % (s + 1, u + 1), frozen)
        return x

    def resnetc5(self, x, deform=True):
        for u in range(self.units[3]):
            x = self._unit(x, self.filter_list[4], 1, u > 0, 'stage4_unit%d' % (u + 1), deform=deform, dilate=not deform)
        return x

    def get_rpn(self, feat, num_anchors):
        c = mx.sym.Convolution(data=feat, kernel=(3, 3), pad=(1, 1), num_filter=512, name='rpn_conv_3x3')
        r = mx.sym.Activation(data=c, act_type='relu', name='rpn_relu')
        cls = mx.sym.Convolution(data=r, kernel=(1, 1), pad=(0, 0), num_filter=2 * num_anchors, name='rpn_cls_score')
        box = mx.sym.Convolution(data=r, kernel=(1, 1), pad=(0, 0), num_filter=4 * num_anchors, name='rpn_bbox_pred')
        return cls, box

    def _trunk(self, cfg, data):
        """-> (C4 features, concat(C4, C5) [fp32])"""
        feat = self.resnetc4(data, fp16=cfg.TRAIN.fp16)
        top = self.resnetc5(feat, deform=True)
        cat = mx.sym.Concat(feat, top, name='cat4')
        if cfg.TRAIN.fp16:
            cat = mx.sym.Cast(data=cat, dtype=np.float32)
        return feat, cat

    def _rpn(self, feat, cat, num_anchors):
        """the RPN reads the concatenated map here (:254); the mask variant reads C4 (resnet_mx_101_e2e_mask.py:291)"""
        return self.get_rpn(cat, num_anchors)

    def _scale(self, cfg):
        return float(cfg.TRAIN.scale) if cfg.TRAIN.fp16 else 1.0

    def _rpn_losses(self, cfg, cls_reshape, bbox_pred, label, target, weight):
        gs = self._scale(cfg)
        prob = mx.sym.SoftmaxOutput(data=cls_reshape, label=label, multi_output=True, normalization='valid', use_ignore=True,
                                    ignore_label=-1, name='rpn_cls_prob', grad_scale=gs)
        l1 = weight * mx.sym.smooth_l1(name='rpn_bbox_loss_', scalar=1.0, data=(bbox_pred - target))
        loss = mx.sym.MakeLoss(name='rpn_bbox_loss', data=l1,
                               grad_scale=3 * gs / float(cfg.TRAIN.BATCH_IMAGES * cfg.TRAIN.RPN_BATCH_SIZE))
        return prob, loss

    def _proposal_kwargs(self, cfg):
        return dict(rpn_pre_nms_top_n=cfg.TEST.RPN_PRE_NMS_TOP_N, rpn_post_nms_top_n=cfg.TEST.RPN_POST_NMS_TOP_N,
                    rpn_min_size=cfg.TEST.RPN_MIN_SIZE, threshold=cfg.TEST.RPN_NMS_THRESH,
                    feature_stride=cfg.network.RPN_FEAT_STRIDE, ratios=tuple(cfg.network.ANCHOR_RATIOS),
                    scales=tuple(cfg.network.ANCHOR_SCALES))

    def _proposal_target(self, cfg, rpn_prob, box, im_info, gt_boxes, valid_ranges):
        """-> (rois, label, bbox_target, bbox_weight, *variant extras)  (:283-284)"""
        return tuple(mx.sym.MultiProposalTarget(cls_prob=rpn_prob, bbox_pred=box, im_info=im_info, gt_boxes=gt_boxes,
                                                valid_ranges=valid_ranges, batch_size=cfg.TRAIN.BATCH_IMAGES,
                                                name='multi_proposal_target'))

    def _extra_train_outputs(self, cfg, feat, extras, grad_scale):
        return []

    def _head(self, feat, rois, num_classes):
        """conv_new_1 features -> offset branch -> deformable PS-RoI pooling -> 2 FC -> cls / bbox (:286-303)."""
        pool = dict(group_size=1, pooled_size=7, sample_per_part=4, part_size=7, output_dim=256, spatial_scale=0.0625)
        t = mx.contrib.sym.DeformablePSROIPooling(name='offset_t', data=feat, rois=rois, no_trans=True, **pool)
        off = mx.sym.FullyConnected(name='offset', data=t, num_hidden=7 * 7 * 2, lr_mult=0.01)
        off = mx.sym.Reshape(data=off, shape=(-1, 2, 7, 7), name='offset_reshape')
        p = mx.contrib.sym.DeformablePSROIPooling(name='deformable_roi_pool', data=feat, rois=rois, trans=off, no_trans=False,
                                                  trans_std=0.1, **pool)
        h = mx.sym.Activation(data=mx.sym.FullyConnected(name='fc_new_1', data=p, num_hidden=1024), act_type='relu',
                              name='fc_new_1_relu')
        h = mx.sym.Activation(data=mx.sym.FullyConnected(name='fc_new_2', data=h, num_hidden=1024), act_type='relu',
                              name='fc_new_2_relu')
        cls = mx.sym.FullyConnected(name='cls_score', data=h, num_hidden=num_classes)
        box = mx.sym.FullyConnected(name='bbox_pred', data=h, num_hidden=4)
        return cls, box

    # ---- graphs ---------------------------------------------------------------------------------
    def get_symbol_rpn(self, cfg, is_train=True):
        A = cfg.network.NUM_ANCHORS
        data = mx.sym.Variable(name='data')
        feat4, cat = self._trunk(cfg, data)
        cls, box = self._rpn(feat4, cat, A)
        cls_r = mx.sym.Reshape(data=cls, shape=(0, 2, -1, 0), name='rpn_cls_score_reshape')
        if is_train:
            prob, loss = self._rpn_losses(cfg, cls_r, box, mx.sym.Variable(name='label'), mx.sym.Variable(name='bbox_target'),
                                          mx.sym.Variable(name='bbox_weight'))
            group = mx.sym.Group([prob, loss])
        else:
            im_info, im_ids = mx.sym.Variable(name='im_info'), mx.sym.Variable(name='im_ids')
            prob = mx.sym.SoftmaxActivation(data=cls_r, mode='channel', name='rpn_cls_prob')
            prob = mx.sym.Reshape(data=prob, shape=(0, 2 * A, -1, 0), name='rpn_cls_prob_reshape')
            rois, scores = mx.sym.MultiProposal(cls_prob=prob, bbox_pred=box, im_info=im_info, name='rois',
                                                batch_size=self.test_nbatch, **self._proposal_kwargs(cfg))
            group = mx.sym.Group([rois, scores, im_ids])
        self.sym = group
        return group

    def get_symbol_rcnn(self, cfg, is_train=True):
        A, C = cfg.network.NUM_ANCHORS, cfg.dataset.NUM_CLASSES
        data = mx.sym.Variable(name='data')
        if is_train:
            label, target, weight = (mx.sym.Variable(name=n) for n in ('label', 'bbox_target', 'bbox_weight'))
            gt_boxes, valid_ranges = mx.sym.Variable(name='gt_boxes'), mx.sym.Variable(name='valid_ranges')
            im_info = mx.sym.Variable(name='im_info')
            scale_label = mx.sym.Variable(name='scale_label') if cfg.TRAIN.AUTO_FOCUS else None
        else:
            im_info, im_ids, chip_ids = (mx.sym.Variable(name=n) for n in ('im_info', 'im_ids', 'chip_ids'))
        feat4, cat = self._trunk(cfg, data)
        cls, box = self._rpn(feat4, cat, A)
        feat = mx.sym.Activation(data=mx.sym.Convolution(data=cat, kernel=(1, 1), num_filter=256, name='conv_new_1'),
                                 act_type='relu', name='conv_new_1_relu')
        focus = None
        if cfg.TRAIN.AUTO_FOCUS or cfg.TEST.AUTO_FOCUS:   # FocusPixel head (:259-267)
            f = mx.sym.Activation(data=mx.sym.Convolution(data=cat, kernel=(3, 3), pad=(1, 1), num_filter=256, name='conv_new_2'),
                                  act_type='relu', name='conv_new_2_relu')
            f = mx.sym.Activation(data=mx.sym.Convolution(data=f, kernel=(1, 1), num_filter=256, name='conv_new_3'),
                                  act_type='relu', name='conv_new_3_relu')
            focus = mx.sym.Convolution(data=f, kernel=(1, 1), num_filter=2, name='conv_new_out')
        cls_r = mx.sym.Reshape(data=cls, shape=(0, 2, -1, 0), name='rpn_cls_score_reshape')
        if is_train:
            gs = self._scale(cfg)
            rpn_prob, rpn_loss = self._rpn_losses(cfg, cls_r, box, label, target, weight)
            pt = self._proposal_target(cfg, rpn_prob, box, im_info, gt_boxes, valid_ranges)
            rois, rlabel, rtarget, rweight = pt[:4]
            rlabel = mx.sym.Reshape(data=rlabel, shape=(-1,), name='label_reshape')
            score, bpred = self._head(feat, rois, C)
            prob = mx.sym.SoftmaxOutput(name='cls_prob', data=score, label=rlabel, normalization='valid', use_ignore=True,
                                        ignore_label=-1, grad_scale=gs)
            outs = [rpn_prob, rpn_loss]
            if cfg.TRAIN.AUTO_FOCUS:
                fr = mx.sym.Reshape(data=focus, shape=(0, 2, -1), name='conv_new_out_reshape')
                outs.append(mx.sym.SoftmaxOutput(name='cls_scale_prob', data=fr, label=scale_label, normalization='valid',
                                                 multi_output=True, use_ignore=True, ignore_label=-1, grad_scale=gs))
            l1 = rweight * mx.sym.smooth_l1(name='bbox_loss_', scalar=1.0, data=(bpred - rtarget))
            bloss = mx.sym.MakeLoss(name='bbox_loss', data=l1, grad_scale=gs / (188.0 * 16.0))
            outs.append(mx.sym.Reshape(data=prob, shape=(cfg.TRAIN.BATCH_IMAGES, -1, C), name='cls_prob_reshape'))
            outs.append(mx.sym.Reshape(data=bloss, shape=(cfg.TRAIN.BATCH_IMAGES, -1, 4), name='bbox_loss_reshape'))
            outs.append(mx.sym.BlockGrad(rlabel))
            outs += self._extra_train_outputs(cfg, feat, pt[4:], gs)
            group = mx.sym.Group(outs)
        else:
            prob = mx.sym.SoftmaxActivation(data=cls_r, mode='channel', name='rpn_cls_prob')
            prob = mx.sym.Reshape(data=prob, shape=(0, 2 * A, -1, 0), name='rpn_cls_prob_reshape')
            rois, _ = mx.sym.MultiProposal(cls_prob=prob, bbox_pred=box, im_info=im_info, name='rois',
                                           batch_size=self.test_nbatch, **self._proposal_kwargs(cfg))
            score, bpred = self._head(feat, rois, C)
            cprob = mx.sym.SoftmaxActivation(name='cls_prob', data=score)
            cprob = mx.sym.Reshape(data=cprob, shape=(self.test_nbatch, -1, C), name='cls_prob_reshape')
            bpred = mx.sym.Reshape(data=bpred, shape=(self.test_nbatch, -1, 4), name='bbox_pred_reshape')
            outs = [rois, cprob, bpred, im_ids]
            if cfg.TEST.AUTO_FOCUS:
                outs.append(mx.sym.SoftmaxActivation(name='scale_prob', data=focus, mode='channel'))
            outs += [im_info, chip_ids]
            group = mx.sym.Group(outs)
        self.sym = group
        return group

    # ---- initialisation of the layers that have no pretrained weights (:450-505) -----------------
    _NEW_RPN = ('rpn_conv_3x3', 'rpn_cls_score', 'rpn_bbox_pred')
    _NEW_RCNN = ('conv_new_1', 'fc_new_1', 'fc_new_2', 'cls_score', 'bbox_pred')
    _NEW_FOCUS = ('conv_new_2', 'conv_new_3', 'conv_new_out')

    def _init(self, arg_params, names, std):
        for n in names:
            shp = self.arg_shape_dict[n + '_weight']
            arg_params[n + '_weight'] = mx.random.normal(0, std, shape=shp) if std > 0 else mx.nd.zeros(shape=shp)
            arg_params[n + '_bias'] = mx.nd.zeros(shape=self.arg_shape_dict[n + '_bias'])

    def init_weight_rpn(self, cfg, arg_params, aux_params):
        self._init(arg_params, ['stage4_unit%d_offset' % u for u in (1, 2, 3)], 0)
        self._init(arg_params, self._NEW_RPN, 0.01)

    def init_weight_rcnn(self, cfg, arg_params, aux_params):
        self.init_weight_rpn(cfg, arg_params, aux_params)
        self._init(arg_params, self._NEW_RCNN, 0.01)
        if cfg.TRAIN.AUTO_FOCUS:
            self._init(arg_params, self._NEW_FOCUS, 0.01)
        self._init(arg_params, ['offset'], 0)

    def init_weight(self, cfg, arg_params, aux_params):
        self.init_weight_rcnn(cfg, arg_params, aux_params)
