"""Debug aid (lives under tests/ because it uses the oracle, which only test code may import): per-node forward values
and output gradients of the MobileNetV2 (C1) step, HIP engine vs oracle.graph_cpu.  Run: python tests/debug_c1_nodes.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['SNIPER_HIP_GRAPHS'] = '0'
from gpu_util import f16r  # noqa: E402
from oracle import graph_cpu  # noqa: E402
from sniper_amd import config as cfgmod  # noqa: E402
from sniper_amd.engine import ops as _ops  # noqa: E402
from sniper_amd.engine.executor import Executor  # noqa: E402
from sniper_amd.symbols.faster import mobilenetv2_e2e as mn  # noqa: E402
from sniper_amd.train import fixed_param_names  # noqa: E402

B, A, F = 2, 15, 16
cfg = cfgmod.mobilenetv2_e2e(batch_images=B)
sym = mn.mobilenetv2_e2e().get_symbol_rcnn(cfg)
shapes = dict(data=(B, 3, 512, 512), valid_ranges=(B, 2), im_info=(B, 3), label=(B, A * F * F),
              bbox_target=(B, 4 * A, F, F), bbox_weight=(B, 4 * A, F, F), gt_boxes=(B, 100, 5), crowd_boxes=(B, 10, 5))
ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym))
rs = np.random.RandomState(11)
args, _, auxs = sym.infer_shape(**shapes)
P, AUX = {}, {}
for name, shp in zip(sym.list_arguments(), args):
    if name in shapes:
        continue
    if name.endswith('_gamma'):
        P[name] = rs.uniform(0.5, 0.9, shp).astype(np.float32)
    elif name.endswith('_beta'):
        P[name] = rs.uniform(0.3, 1.2, shp).astype(np.float32)
    elif name.endswith('_bias'):
        P[name] = np.zeros(shp, np.float32)
    elif name.startswith('offset'):
        P[name] = (rs.standard_normal(shp) * 1e-3).astype(np.float32)
    elif any(name.startswith(h) for h in ('rpn_', 'conv_new_1', 'fc_new', 'cls_score', 'bbox_pred')):
        P[name] = (rs.standard_normal(shp) * 0.01).astype(np.float32)
    else:
        P[name] = (rs.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[1:]))).astype(np.float32)
for name, shp in zip(sym.list_auxiliary_states(), auxs):
    AUX[name] = np.ones(shp, np.float32) if name.endswith('_var') else np.zeros(shp, np.float32)
P = {k: (f16r(v) if v.ndim > 1 else v) for k, v in P.items()}
ex.set_params(P, AUX)
gt = -np.ones((B, 100, 5), np.float32)
for b in range(B):
    n = 40
    c = rs.uniform(60, 450, (n, 2))
    wh = rs.uniform(40, 320, (n, 2))
    gt[b, :n, :4] = np.clip(np.concatenate((c - wh / 2, c + wh / 2), 1), 0, 511)
    gt[b, :n, 4] = rs.randint(1, 81, n)
inp = dict(data=f16r(rs.standard_normal((B, 3, 512, 512)) * 50),   # fp16-representable pixels: the stem packs the image to fp16
           valid_ranges=np.array([[0, 512]] * B, np.float32), im_info=np.array([[512, 512, 1.0]] * B, np.float32),
           label=rs.choice([-1, 0, 1], size=(B, A * F * F), p=[0.9, 0.07, 0.03]).astype(np.float32),
           bbox_target=(rs.standard_normal((B, 4 * A, F, F)) * 0.3).astype(np.float32),
           bbox_weight=(rs.uniform(size=(B, 4 * A, F, F)) < 0.05).astype(np.float32), gt_boxes=gt,
           crowd_boxes=-np.ones((B, 10, 5), np.float32))

# record every step's output value and incoming gradient on the device
rec = {}
for st in ex.steps:
    orig = st.backward

    def wrapped(st=st, orig=orig):
        y = getattr(st, 'y', None)
        if y is not None and getattr(y, 'grad', None) is not None and y.t is not None:
            def to_ref(t, v=y):
                t = t.float()
                if v.fmt == 'act' and len(v.shape) == 4:
                    t = t.permute(0, 3, 1, 2)
                return t.reshape(v.shape).cpu().numpy()
            rec[st.node.name] = (to_ref(y.t), to_ref(y.grad))
        return orig()
    st.backward = wrapped
outs = ex.forward(inp, is_train=True)
ex.backward()
torch.cuda.synchronize()
mpt = [n for n in sym._topo() if n.op == 'MultiProposalTarget'][0]
ov = {(mpt.name, i): ex.vals[(id(mpt), i)].t.cpu().numpy().reshape(ex.vals[(id(mpt), i)].shape) for i in range(4)}
names = list(rec.keys())
force = {}
for st in ex.steps:
    if st.node.name in rec and not (type(st).__name__ == 'BatchNormStep' and getattr(st, 'act', 0)) and \
            getattr(st, 'fused_residual', None) is None:
        force[st.node.name] = rec[st.node.name][0]
want, wgrads, probes = graph_cpu.run(sym, P, AUX, inp, overrides=ov, fork_ops=False, fp16_storage=True, probe=names, force=force)
le = graph_cpu.run.local_err
print('teacher-forced per-node forward mismatch: max %.5f at %s; nodes > 2e-3: %s' % (
    max(le.values()), max(le, key=le.get), [(k, round(v, 5)) for k, v in le.items() if v > 2e-3]))
for name, p_ in ex.params.items():
    if p_.trainable:
        g, w = p_.to_reference(p_.grad.detach().cpu().numpy()), wgrads[name]
        print('PARAM %-44s relL2 %.4f' % (name, float(np.linalg.norm(g.astype(np.float64) - w) / (np.linalg.norm(w) + 1e-20))))


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-20))


print('%-40s %10s %10s %12s' % ('node (backward order)', 'value relL2', 'grad relL2', 'max|grad|'))
for k in names:
    if k not in probes or probes[k][1] is None:
        continue
    v, g = probes[k]
    gv, gg = rec[k]
    if gv.shape != v.shape:
        continue
    print('%-40s %10.4f %10.4f %12.4g' % (k, rel(gv, v), rel(gg, g), np.abs(g).max()))
