"""Helper of tests/test_gpu_inference.py::test_rank_sharded_inference_equals_one_process -- one rank of a 2-rank gloo group sharing
one GPU (a rehearsal of the control flow, not a measurement): rank 0 also runs the whole roidb alone and compares."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import sniper_amd.mx as mx
    from sniper_amd import config as cfgmod
    from sniper_amd import inference
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)

    class Imdb(object):
        num_classes, classes, name, result_path = 81, None, 'synthetic', None
    rs = np.random.RandomState(5)
    base = [{'image': rs.randint(0, 256, (240, 320, 3)).astype(np.uint8), 'width': 320, 'height': 240, 'flipped': False,
             'gt_overlaps': np.zeros((1, 81), np.float32)} for _ in range(5)]
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.SCALES = ((240, 320), (480, 640))
    cfg.TEST.BATCH_IMAGES = (2, 2)
    cfg.TEST.VALID_RANGES = ((40, -1), (-1, 60))
    cfg.TEST.DO_PRUNING = (False, True)
    cfg.TEST.CHIP_HYPERPARAMS = ((3, 0.3, 4), (-1, -1, -1))
    cfg.TEST.MAX_PER_IMAGE = 50
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 500, 100

    def fmap(scale_i, image, chip, net_map):          # keyed on the GLOBAL image index: odd images get two FocusChips
        out = np.zeros_like(np.asarray(net_map, np.float32))
        out[0] = 1.0
        out[1, 1:3, 1:3] = 0.9
        if image % 2:
            out[1, -3:-1, -3:-1] = 0.9
        out[0] -= out[1]
        return out
    cache = {}
    got = inference.imdb_detection_wrapper(rn.resnet_mx_101_e2e, cfg, Imdb(), [dict(r) for r in base], [mx.gpu(0)], None, None,
                                           module_cache=cache, focus_map_fn=fmap)          # rank / world from the process group
    ok = True
    if rank == 0:
        want = inference.imdb_detection_wrapper(rn.resnet_mx_101_e2e, cfg, Imdb(), [dict(r) for r in base], [mx.gpu(0)], None, None,
                                                module_cache=cache, focus_map_fn=fmap, rank=0, world=1)
        n = 0
        for j in range(1, 81):
            for wi, gi in zip(want[j], got[j]):
                ok = ok and np.array_equal(np.asarray(wi), np.asarray(gi))
                n += len(wi)
        ok = ok and n > 0
        print('SHARD_RESULT ok=%d boxes=%d' % (int(ok), n), flush=True)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
