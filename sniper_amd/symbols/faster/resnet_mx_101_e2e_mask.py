"""ResNet-101 SNIPER with the auxiliary mask branch (configs/faster/sniper_res101_e2e_mask*.yml).

Same graph, parameter names and outputs as the reference's symbols/faster/resnet_mx_101_e2e_mask.py; the differences from
the plain R101 network (sniper_amd/symbols/faster/resnet_mx_101_e2e.py) are

  * the RPN reads the C4 features (cast to fp32) instead of concat(C4, C5)                      (:153-156, :291)
  * MultiProposalTargetMask also returns the 50 mask RoIs per chip and the GT row each one matched    (:317-318)
  * mask head (training graph only; a training signal, the test graph is the detector's):            (:238-254, :374-405)
      14x14 deformable PS-RoI pooling (+ its zero-initialised offset FC) -> 4 x [3x3 conv 256 + relu] -> 2x2/2
      deconvolution 256 + relu -> 1x1 conv to 2 x 80 maps at 28x28; MaskRcnnTarget rasterises the matched GT polygon
      into the RoI's 28x28 grid; `pick` takes the RoI class's negative / positive map -> 2-way per-pixel softmax.
"""
import numpy as np

import sniper_amd.mx as mx

from . import resnet_mx_101_e2e as base

checkpoint_callback = base.checkpoint_callback

MASK_SIZE, MASK_ROIS, MASK_CLASSES = 28, 50, 80


class resnet_mx_101_e2e_mask(base.resnet_mx_101_e2e):
    _NEW_MASK = tuple('mask_conv_3x3_%d' % (i + 1) for i in range(4)) + ('mask_out',)

    def get_rpn(self, feat, num_anchors):
        return base.resnet_mx_101_e2e.get_rpn(self, mx.sym.Cast(data=feat, dtype=np.float32), num_anchors)

    def _rpn(self, feat, cat, num_anchors):
        return self.get_rpn(feat, num_anchors)

    def _proposal_target(self, cfg, rpn_prob, box, im_info, gt_boxes, valid_ranges):
        return tuple(mx.sym.MultiProposalTargetMask(cls_prob=rpn_prob, bbox_pred=box, im_info=im_info, gt_boxes=gt_boxes,
                                                    valid_ranges=valid_ranges, batch_size=cfg.TRAIN.BATCH_IMAGES,
                                                    name='multi_proposal_target_mask'))

    def get_mask_head(self, feat, num_layers=4, num_classes=MASK_CLASSES):
        x = feat
        for i in range(num_layers):
            x = mx.sym.Convolution(data=x, kernel=(3, 3), pad=(1, 1), num_filter=256, name='mask_conv_3x3_%d' % (i + 1))
            x = mx.sym.Activation(data=x, act_type='relu', name='mask_relu_%d' % (i + 1))
        x = mx.sym.Cast(data=x, dtype=np.float32)
        x = mx.sym.Deconvolution(data=x, kernel=(2, 2), stride=(2, 2), pad=(0, 0), num_filter=256, name='mask_deconv')
        x = mx.sym.Activation(data=x, act_type='relu', name='mask_deconv_relu')
        return mx.sym.Convolution(data=x, kernel=(1, 1), pad=(0, 0), num_filter=num_classes * 2, name='mask_out')

    def _extra_train_outputs(self, cfg, feat, extras, grad_scale):
        mask_rois, mask_ids = extras
        gt_masks = mx.sym.Variable(name='gt_masks')
        pool = dict(group_size=1, pooled_size=14, sample_per_part=4, part_size=14, output_dim=256, spatial_scale=0.0625)
        t = mx.contrib.sym.DeformablePSROIPooling(name='mask_offset_t', data=feat, rois=mask_rois, no_trans=True, **pool)
        off = mx.sym.FullyConnected(name='mask_offset', data=t, num_hidden=14 * 14 * 2, lr_mult=0.01)
        off = mx.sym.Reshape(data=off, shape=(-1, 2, 14, 14), name='mask_offset_reshape')
        p = mx.contrib.sym.DeformablePSROIPooling(name='mask_deformable_roi_pool', data=feat, rois=mask_rois, trans=off,
                                                  no_trans=False, trans_std=0.1, **pool)
        pred = self.get_mask_head(mx.sym.Cast(data=p, dtype=np.float16))
        targets, ncls = mx.sym.MaskRcnnTarget(rois=mask_rois, mask_polys=gt_masks, mask_ids=mask_ids,
                                              batch_size=cfg.TRAIN.BATCH_IMAGES, mask_size=MASK_SIZE, num_proposals=MASK_ROIS,
                                              max_polygon_len=500, max_num_gts=100, num_classes=MASK_CLASSES)
        pcls = ncls + MASK_CLASSES
        pos = mx.sym.pick(pred, index=pcls, axis=1, keepdims=True)
        neg = mx.sym.pick(pred, index=ncls, axis=1, keepdims=True)
        both = mx.sym.Concat(neg, pos, name='pred_cat')
        prob = mx.sym.SoftmaxOutput(data=both, label=targets, multi_output=True, normalization='valid', use_ignore=True,
                                    ignore_label=-1, name='mask_cls_prob', grad_scale=grad_scale)
        return [prob, mx.sym.BlockGrad(targets)]

    def init_weight_mask(self, cfg, arg_params, aux_params):
        for n in self._NEW_MASK:
            arg_params[n + '_weight'] = mx.random.normal(0, 0.01, shape=self.arg_shape_dict[n + '_weight'])
            arg_params[n + '_bias'] = mx.nd.zeros(shape=self.arg_shape_dict[n + '_bias'])
        arg_params['mask_deconv_weight'] = mx.random.normal(0, 0.01, shape=self.arg_shape_dict['mask_deconv_weight'])
        self._init(arg_params, ['mask_offset'], 0)

    def init_weight_rcnn(self, cfg, arg_params, aux_params):
        base.resnet_mx_101_e2e.init_weight_rcnn(self, cfg, arg_params, aux_params)
        if 'mask_out_weight' in self.arg_shape_dict:      # the test graph has no mask head
            self.init_weight_mask(cfg, arg_params, aux_params)
