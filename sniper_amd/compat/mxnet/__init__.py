"""`import mxnet` -> sniper_amd.mx.  Put `<repo>/sniper_amd/compat` on sys.path and
the reference's files (`import mxnet as mx`) bind to the MI355X engine."""
import sys

import sniper_amd.mx as _mx

_mx.alias_as('mxnet')
sys.modules[__name__] = _mx
