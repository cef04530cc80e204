"""HBM a bound test-time executor holds, stage by stage (bind, parameters, first eager forward, capture, replay), for a list of
batch shapes of ONE Module -- what the shared activation pool (engine/executor.py::ActivationPool) leaves per further shape.

    python tools/exe_memory.py [nbatch] [HxW ...]          default: 2 1408x2048 704x1024 1024x1408
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gb():
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9


def main():
    import sniper_amd.mx as mx
    from sniper_amd import config as cfgmod
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    sizes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[2:]] or [(1408, 2048), (704, 1024), (1024, 1408)]
    cfg = cfgmod.res101_e2e_autofocus()
    bind = [('data', (nb, 3) + sizes[0]), ('im_info', (nb, 3)), ('im_ids', (nb,)), ('chip_ids', (nb,))]
    net = rn.resnet_mx_101_e2e(n_proposals=400, test_nbatch=nb)
    sym = net.get_symbol_rcnn(cfg, is_train=False)
    mod = mx.mod.Module(symbol=sym, context=[mx.gpu(0)], data_names=[k for k, _ in bind], label_names=None)
    print('start                      allocated %.2f GB reserved %.2f GB' % gb())
    mod.bind(bind, None, for_training=False)
    print('bound %s          allocated %.2f GB reserved %.2f GB' % ((sizes[0],) + gb()))
    mod.init_params(arg_params=None, aux_params=None, allow_missing=True)
    print('parameters                 allocated %.2f GB reserved %.2f GB' % gb())
    rs = np.random.RandomState(0)
    for h, w in sizes:
        shp = [('data', (nb, 3, h, w))] + bind[1:]
        data = [mx.nd.array((rs.standard_normal((nb, 3, h, w)) * 40).astype(np.float32)),
                mx.nd.array(np.tile(np.array([[h, w, 1.0]], np.float32), (nb, 1))), mx.nd.array(np.arange(nb, dtype=np.float32)),
                mx.nd.array(np.zeros(nb, np.float32))]
        batch = mx.io.DataBatch(data=data, label=None, pad=0, index=None, provide_data=shp, provide_label=None)
        for call in range(3):
            mod.forward(batch, is_train=False)
            [o.asnumpy() for o in mod.get_outputs()]
            print('%s forward %d (%s)   allocated %.2f GB reserved %.2f GB' % (
                (h, w), call, ('eager', 'capture', 'replay')[call], *gb()))
        pool = getattr(mod, '_act_pool', None)
        ws = sum(e.ws.buf.numel() for e in mod._exes.values() if e.ws.buf is not None)
        print('    pool %.2f GB in %d buffers; executors\' workspaces %.3f GB' % (
            (pool.nbytes() / 1e9 if pool else 0.0), (len(pool.buffers) if pool else 0), ws / 1e9))

if __name__ == '__main__':
    main()
