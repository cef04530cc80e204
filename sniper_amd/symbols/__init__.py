"""Network definitions with the reference's symbol-class contract: module
``symbols.faster.<name>`` exposing class ``<name>(n_proposals, momentum, fix_bn, test_nbatch)`` with
``get_symbol_rcnn / get_symbol_rpn(cfg, is_train)``, ``infer_shape``, ``init_weight_rcnn / rpn``,
``get_bbox_param_names`` and a module-level ``checkpoint_callback``."""
