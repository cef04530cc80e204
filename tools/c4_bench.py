"""BASELINE configs[3] (C4) at ONE GPU's share: ResNet-101 SNIPER with the position-sensitive R-FCN head
(sniper_amd/symbols/faster/resnet_mx_101_e2e_rfcn.py: group_size 7 deformable PS-RoI pooling on 7*7*81 / 7*7*4 maps, bin vote), 16 chips
of 512 x 512 per GPU (batch 128 over 8 GPUs; SURVEY 8(d) C4, spec ours: the reference's master branch only pools with group_size 1,
symbols/faster/resnet_mx_101_e2e.py:286-293).  Same step as bench.py -- GPU anchor labelling + forward + backward + SGD on
HBM-resident synthetic chips, hipGraph replay -- then the head alone: HIP events around every head operator of two eager steps
(rfcn_* convolutions, the PS-RoI pooling calls forward + both gradients, the bin vote).  Prints one JSON object.

    python tools/c4_bench.py [steps (20)] [warmup (5)] [chips (16)]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HEAD_PREFIXES = ('rfcn_', 'psroipooled_', 'ave_')


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    chips = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    import torch
    import sniper_amd.mx as mx
    from sniper_amd import config as cfgmod
    from sniper_amd.train import Trainer
    cfg = cfgmod.res101_e2e(batch_images=chips)
    cfg.symbol = 'resnet_mx_101_e2e_rfcn'
    tr = Trainer(batch_images=chips, n_images=48, seed=0, rank_local=True, cfg=cfg)
    batches = [tr.batch] + [tr.next_batch() for _ in range(3)]
    anchors = tr.iter.anchors
    packed = [anchors.pack_device(b.worker_data) for b in batches]

    def step(i):
        b = batches[i % len(batches)]
        lab = anchors.assign(packed=packed[i % len(batches)], seed=i)
        label = [mx.nd.NDArray(lab['label']), mx.nd.NDArray(lab['bbox_target']), mx.nd.NDArray(lab['bbox_weight']), mx.nd.NDArray(lab['gt_boxes'])]
        tr.step(mx.io.DataBatch(data=b.data, label=label, pad=0, index=None, provide_data=b.provide_data, provide_label=b.provide_label))

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ex = tr.mod.exe
    graphs = bool(ex.use_graphs and ex._graph_fb is not None)
    # ---- the head's own time: events around its operators, eager steps (a replayed graph cannot be bracketed)
    saved = (ex.use_graphs, ex._graph_fb, ex._graph_up)
    ex.use_graphs, ex._graph_fb, ex._graph_up = False, None, None
    head = [s for s in ex.steps if s.node.name.startswith(HEAD_PREFIXES)]
    rec = []

    def wrap(s, what):
        fn = getattr(s, what)

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            rec.append((s.node.name, what, e0, e1))
            return r
        setattr(s, what, timed)
        return fn
    step(0)
    orig = [(s, w, wrap(s, w)) for s in head for w in ('forward', 'backward')]
    n_prof = 2
    for i in range(n_prof):
        step(i)
    torch.cuda.synchronize()
    for s, w, fn in orig:
        setattr(s, w, fn)
    ex.use_graphs, ex._graph_fb, ex._graph_up = saved
    by = {}
    for name, what, e0, e1 in rec:
        k = '%s %s' % (name, what)
        by[k] = by.get(k, 0.0) + e0.elapsed_time(e1) / n_prof
    # (weight gradients of the head's 1 x 1 convolutions are queued into the step's batched weight-gradient tables and are not in
    #  these brackets; the brackets include their launch gaps: an upper bound of the kernels' own time)
    head_ms = sum(by.values())
    out = {'what': 'BASELINE configs[3] (C4) at one GPU\'s share: R101 SNIPER + position-sensitive R-FCN head (group_size 7 PS-RoI pooling, '
                   'spec ours), %d chips x 512x512 per GPU, synthetic; step = anchor labelling + fwd + bwd + SGD, hipGraph replay' % chips,
           'value': round(chips * steps / dt, 2), 'unit': 'chips/s', 'ms_per_step': round(dt / steps * 1e3, 3), 'chips_per_gpu': chips,
           'steps': steps, 'warmup': warmup, 'graphs': graphs, 'head_ms': round(head_ms, 3),
           'head_operators': len(head), 'head_by_operator_ms': {k: round(v, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]}}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
