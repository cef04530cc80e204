"""SURVEY 8(e), last sentence: inference shards images over ranks and gathers the detections on rank 0 before `aggregate`
(the reference does the same across its forked jobs, lib/inference.py:456-500).  CPU, world 2 and 3 over gloo: the per-rank forward
passes are replaced by a deterministic stand-in keyed on the image (the kernels need a GPU; tests/test_gpu_inference.py runs the real
ones on two ranks sharing one card) -- what is exercised is the sharding, the local -> global image mapping of the FocusPixel-map
callback, the object gather, the merge and that rank 0 alone aggregates."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Imdb(object):
    num_classes, classes, name, result_path = 4, None, 'fake', None


class _Cfg(object):
    class TEST(object):
        SCALES = [(480, 512), (800, 1280)]
        BATCH_IMAGES = [8, 2]
        VALID_RANGES = [(-1, -1), (-1, -1)]
        NMS, NMS_SIGMA = -1, 0.55
        MAX_PER_IMAGE = 100


def _fake_scale_detections(inference, roidb, focus_map_fn):
    """stand-in of _multi_scale_detections: per scale, [class][local image][chip] -> (n, 5) rows drawn from the image's own id;
    every second chip comes back in the `compact` form of the GPU path.  Calls the map callback like the real pass does."""
    out = []
    for s_i in range(2):
        d = inference._Detections([[[] for _ in roidb] for _ in range(_Imdb.num_classes)])
        for li, r in enumerate(roidb):
            n_chips = 1 + (r['id'] + s_i) % 3
            if focus_map_fn is not None:
                m = focus_map_fn(s_i, li, 0, np.zeros((2, 3, 3), np.float32))
                assert int(m[0, 0, 0]) == r['id'], 'the map callback must see the GLOBAL image index'
            for j in range(_Imdb.num_classes):
                d[j][li] = [None] * n_chips
            for c in range(n_chips):
                rs = np.random.RandomState(1000 * r['id'] + 10 * s_i + c)
                lens = rs.randint(0, 4, _Imdb.num_classes - 1)
                big = rs.uniform(0, 100, (int(lens.sum()), 5))
                ends = np.cumsum(lens)
                for j in range(1, _Imdb.num_classes):
                    d[j][li][c] = big[ends[j - 1] - lens[j - 1]:ends[j - 1]]
                d[0][li][c] = np.zeros((0, 5))
                if c % 2 == 0:
                    d.compact[(li, c)] = (big, lens)
        out.append(d)
    return out


def _canon(dets):
    return [([[[np.asarray(x).tolist() for x in img] for img in cls] for cls in d],
             sorted((k, v[0].tolist(), v[1].tolist()) for k, v in d.compact.items())) for d in dets]


def _run(rank, world, port, n_images, out):
    import torch.distributed as dist
    from sniper_amd import inference
    if world > 1:
        os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    calls = {'aggregate': 0}
    inference._multi_scale_detections = lambda sym, cfg, imdb, roidb, ctx, a, b, vis, cache, fmap, jobs, lanes, rows=None: \
        _fake_scale_detections(inference, roidb, fmap)

    def aggregate(self, scale_cls_dets, **kw):
        calls['aggregate'] += 1
        return 'aggregated %d images' % self.num_images
    inference.Tester.aggregate = aggregate
    roidb = [{'id': g, 'width': 64, 'height': 48} for g in range(n_images)]
    fmap = lambda s_i, g, c, m: np.full_like(m, g)          # noqa: E731
    res, dets = inference.imdb_detection_wrapper(None, _Cfg, _Imdb(), roidb, None, None, None, focus_map_fn=fmap, return_scale_dets=True)
    if rank == 0:
        assert res == 'aggregated %d images' % n_images and calls['aggregate'] == 1
        out['dets'] = _canon(dets)
    else:
        assert res is None and dets is None and calls['aggregate'] == 0
    out[rank] = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _gathered(world, n_images):
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, n_images, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, 'rank exited with %s' % p.exitcode
    assert all(out.get(r) for r in range(world))
    return out['dets']


def test_sharded_inference_gathers_the_single_process_detections():
    single = {}
    _run(0, 1, 0, 11, single)
    for world, n in ((2, 11), (3, 11)):
        assert _gathered(world, n) == single['dets'], world
    # fewer images than ranks: a rank with nothing to do still takes part in the gather
    one = {}
    _run(0, 1, 0, 1, one)
    assert _gathered(2, 1) == one['dets']


def test_shard_images_is_a_partition():
    from sniper_amd.inference import shard_images
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            parts = [shard_images(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_default_lanes(monkeypatch):
    """SNIPER_LANES wins; without it the choice follows free HBM (3 lanes with >= 96 GiB free, else 1); no GPU -> 1."""
    import torch
    from sniper_amd import inference
    monkeypatch.setenv('SNIPER_LANES', '2')
    assert inference.default_lanes() == 2
    monkeypatch.setenv('SNIPER_LANES', '0')
    assert inference.default_lanes() == 1
    monkeypatch.delenv('SNIPER_LANES')
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    assert inference.default_lanes() == 1
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda *a: (200 << 30, 288 << 30))
    assert inference.default_lanes() == 3
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda *a: (40 << 30, 288 << 30))
    assert inference.default_lanes() == 1
