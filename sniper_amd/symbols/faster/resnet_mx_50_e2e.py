"""ResNet-50 SNIPER (configs/faster/sniper_res50_e2e.yml): the reference's symbols/faster/resnet_mx_50_e2e.py is its
ResNet-101 file with `units = (3, 4, 6, 3)` (:26) and without the AutoFocus head; same trunk builder, heads, losses and
parameter names here (sniper_amd/symbols/faster/resnet_mx_101_e2e.py)."""
from . import resnet_mx_101_e2e as base

checkpoint_callback = base.checkpoint_callback


class resnet_mx_50_e2e(base.resnet_mx_101_e2e):
    UNITS = (3, 4, 6, 3)
