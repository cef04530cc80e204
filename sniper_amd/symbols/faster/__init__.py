from . import resnet_mx_101_e2e  # noqa: F401
