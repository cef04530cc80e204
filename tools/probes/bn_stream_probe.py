"""sn_bn_apply / bn_bwd_dx back to back on the stage-3 maps against torch's copy on the same buffers (same cache state: the two
tensors of a launch fit the Infinity Cache).   python tools/probes/bn_stream_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from sniper_amd import hip
d = torch.device('cuda', 0)


def us(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


M = 20480
for C in (256, 1024, 2048):
    x = torch.randn(M, C, device=d).half()
    y = torch.empty_like(x)
    dy, acc = torch.randn(M, C, device=d).half(), torch.randn(M, C, device=d).half()
    sc, sh = torch.rand(C, device=d) + 0.5, torch.randn(C, device=d)
    mean, inv = torch.randn(C, device=d) * 0.1, torch.rand(C, device=d) + 0.5
    ws = torch.zeros(hip.query('sn_bn_workspace_bytes', M, C), dtype=torch.uint8, device=d)
    dg, db = torch.zeros(C, device=d), torch.zeros(C, device=d)
    with torch.cuda.stream(torch.cuda.Stream()):
        pass
    st = hip.stream()
    t_apply = us(lambda: hip.call('sn_bn_apply', x, y, M, C, C, C, sc, sh, 1, st))
    t_copy = us(lambda: y.copy_(x))
    t_relu = us(lambda: torch.relu(x, out=y) if False else torch.clamp_min(x, 0, out=y))
    t_bwd = us(lambda: hip.call('sn_bn_backward', dy, x, acc, y, M, C, C, C, C, C, sc, sh, mean, inv, 1, ws, dg, db, st))
    t_add3 = us(lambda: torch.add(torch.add(dy, x, out=y), acc, out=y))
    mb = M * C * 2 / 1e6
    print('C %4d (%5.1f MB per tensor): sn_bn_apply %.1f us   torch copy %.1f   torch clamp_min %.1f   |   sn_bn_backward (reduce + finalize + dx) %.1f   torch 2 adds %.1f'
          % (C, mb, t_apply, t_copy, t_relu, t_bwd, t_add3), flush=True)
