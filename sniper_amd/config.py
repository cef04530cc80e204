"""Configuration presets for the hot path.

The reference keeps its settings in a global ``easydict`` (``configs/faster/default_configs.py``)
overlaid by YAML files.  Those files stay the reference's; what lives here are the *values* the
hot path consumes for the BASELINE configs (``configs/faster/sniper_res101_e2e.yml``,
``sniper_mobilenetv2_e2e.yml``), as an attribute dict with the same key names, so that the host
mirrors (`sniper_amd.data`, `sniper_amd.symbols`) can be driven either by these presets or by the
reference's own ``config`` object when its checkout is on ``sys.path``.
"""
import numpy as np


class AttrDict(dict):
    """Minimal stand-in for easydict.EasyDict (not installed here): attribute access over a dict,
    nested dicts converted recursively, py2 ``has_key`` kept because default_configs.py:212 uses it."""

    def __init__(self, d=None, **kw):
        super(AttrDict, self).__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super(AttrDict, self).__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__

    def has_key(self, k):
        return k in self


def _base():
    c = AttrDict()
    c.CLASS_AGNOSTIC = True
    c.default = {'kvstore': 'device'}
    c.network = {
        'PIXEL_MEANS': np.array([103.939, 116.779, 123.68]),
        'RPN_FEAT_STRIDE': 16,
        'FIXED_PARAMS': ['conv0', 'bn0', 'stage1'],
        'ANCHOR_RATIOS': (0.5, 1, 2),
        'ANCHOR_SCALES': (2, 4, 7, 10, 13, 16, 24),
        'NUM_ANCHORS': 21,
    }
    c.dataset = {'NUM_CLASSES': 81, 'dataset': 'coco'}
    c.TRAIN = {
        'ONLY_PROPOSAL': False, 'CPP_CHIPS': True, 'USE_NEG_CHIPS': True, 'CHIPS_DB_PARTS': 20,
        'WITH_MASK': False, 'AUTO_FOCUS': False, 'AUTO_FOCUS_SMALL_THRESH': -1, 'AUTO_FOCUS_DC_LOW': -1,
        'AUTO_FOCUS_DC_HIGH': -1,
        'SCALES': ((1400, 2000), (800, 1280), (-1, 512)),
        'VALID_RANGES': ((-1, 80), (32, 150), (120, -1)),
        'lr': 0.015, 'lr_step': '5.33', 'lr_factor': 0.1, 'warmup': True, 'warmup_lr': 0.0005, 'warmup_step': 1000,
        'wd': 0.0001, 'momentum': 0.9, 'fp16': True, 'scale': 100.0, 'begin_epoch': 0, 'end_epoch': 7,
        'BATCH_IMAGES': 16, 'NUM_PROCESS': 64, 'NUM_THREAD': 8,
        'FG_THRESH': 0.5, 'BG_THRESH_HI': 0.5, 'BG_THRESH_LO': 0.0,
        'RPN_BATCH_SIZE': 256, 'RPN_FG_FRACTION': 0.5, 'RPN_POSITIVE_OVERLAP': 0.5, 'RPN_NEGATIVE_OVERLAP': 0.4,
        'RPN_NMS_THRESH': 0.7, 'RPN_PRE_NMS_TOP_N': 6000, 'RPN_POST_NMS_TOP_N': 300, 'RPN_MIN_SIZE': 0,
        'BBOX_MEANS': (0.0, 0.0, 0.0, 0.0), 'BBOX_STDS': (0.1, 0.1, 0.2, 0.2),
    }
    c.TEST = {
        'SCALES': ((1400, 2000), (800, 1280), (480, 512)), 'VALID_RANGES': ((-1, 90), (32, 180), (75, -1)),
        'BATCH_IMAGES': (2, 2, 4), 'RPN_NMS_THRESH': 0.7, 'RPN_PRE_NMS_TOP_N': 6000, 'RPN_POST_NMS_TOP_N': 300,
        'RPN_MIN_SIZE': 0, 'NMS': -1, 'NMS_SIGMA': 0.55, 'MAX_PER_IMAGE': 200, 'AUTO_FOCUS': False,
        'DO_PRUNING': (False, False, False), 'CHIP_HYPERPARAMS': ((-1, -1, -1),) * 3, 'CONCURRENT_JOBS': 1,
    }
    return c


def res101_e2e(batch_images=20):
    """configs/faster/sniper_res101_e2e.yml (BASELINE C2/C3; BATCH_IMAGES 20 per BASELINE.json)."""
    c = _base()
    c.symbol = 'resnet_mx_101_e2e'
    c.TRAIN.BATCH_IMAGES = batch_images
    return c


def res101_e2e_mask(batch_images=20):
    """configs/faster/sniper_res101_e2e_mask.yml: the R101 detector trained with the auxiliary mask branch."""
    c = res101_e2e(batch_images)
    c.symbol = 'resnet_mx_101_e2e_mask'
    c.TRAIN.WITH_MASK = True
    return c


def res101_e2e_autofocus(batch_images=20):
    """configs/faster/sniper_res101_e2e_autofocus.yml TEST section (BASELINE C5): coarse-to-fine scales, FocusChips."""
    c = res101_e2e(batch_images)
    c.TEST.SCALES = ((480, 512), (800, 1280), (1400, 2000))
    c.TEST.BATCH_IMAGES = (8, 8, 2)
    c.TEST.VALID_RANGES = ((75, -1), (32, 180), (-1, 75))
    c.TEST.AUTO_FOCUS = True
    c.TEST.DO_PRUNING = (False, True, True)
    c.TEST.CHIP_HYPERPARAMS = ((3, 0.02, 16), (3, 0.2, 20), (-1, -1, -1))
    c.TEST.USE_CACHE = (False, False, False)
    c.TEST.CONCURRENT_JOBS = 1
    return c


def mobilenetv2_e2e(batch_images=2):
    """configs/faster/sniper_mobilenetv2_e2e.yml, reduced to one scale for BASELINE C1."""
    c = _base()
    c.symbol = 'mobilenetv2_e2e'
    c.network.RPN_FEAT_STRIDE = 32
    c.network.ANCHOR_SCALES = (1, 2, 4, 8, 12)
    c.network.NUM_ANCHORS = 15
    c.network.FIXED_PARAMS = ['conv1', 'bn_conv1', 'res2', 'bn2', 'res3', 'bn3', 'res4', 'bn4', 'gamma', 'beta']
    c.TRAIN.SCALES = ((-1, 512),)
    c.TRAIN.VALID_RANGES = ((-1, -1),)
    c.TRAIN.BATCH_IMAGES = batch_images
    return c
