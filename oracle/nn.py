"""CPU restatements of the network operators whose source lives in the un-vendored SNIPER-mxnet
fork (TEST INFRASTRUCTURE ONLY).  **Parity unpinned**: the reference tree holds neither the
operator sources nor any test vector for them; these follow the published
definitions (Faster R-CNN proposal layer, Deformable ConvNets v1) and, where those are silent, the
choices documented in DESIGN.md.  Standard ops (conv, BN, softmax) are cross-checked against
torch-CPU fp32 in tests/.  numpy, loops -> small sizes only.
"""
import numpy as np

from . import capi
from .data_path import generate_anchors


# ---------------------------------------------------------------------------------------------
# MultiProposal / MultiProposalTarget (symbols/faster/resnet_mx_101_e2e.py:283-284, 347-355)
# ---------------------------------------------------------------------------------------------
def proposals(cls_prob, bbox_pred, im_info, feat_stride, scales, ratios, pre_nms, post_nms, nms_thresh, min_size=0):
    """cls_prob (B,2,A*F,F), bbox_pred (B,4A,F,F) float32.  Returns rois (B*post,5), scores (B*post,)
    and per image the sorted pre-NMS boxes + kept indices (for set-level checks).
    Steps: anchors (generate_anchor.py) in (y, x, a) order; float32 decode = nonlinear_pred
    (bbox_transform.py:93-130); clip to (im_h-1, im_w-1); boxes below min_size*scale get score -1;
    stable sort by descending score; top pre_nms; NMS (nms.py:90-127 semantics); first post_nms,
    cyclically repeated when fewer survive."""
    B = cls_prob.shape[0]
    A, Fh, Fw = bbox_pred.shape[1] // 4, bbox_pred.shape[2], bbox_pred.shape[3]     # test images are not square
    base = generate_anchors(feat_stride, ratios, np.array(scales, np.float32)).astype(np.float32)
    rois = np.zeros((B * post_nms, 5), np.float32)
    scores_out = np.zeros((B * post_nms,), np.float32)
    dbg = []
    f32 = np.float32
    for b in range(B):
        fg = cls_prob[b].reshape(2, A, Fh, Fw)[1].transpose(1, 2, 0).reshape(-1)  # (y,x,a)
        d = bbox_pred[b].reshape(A, 4, Fh, Fw).transpose(2, 3, 0, 1).reshape(-1, 4).astype(f32)
        sx, sy = np.meshgrid(np.arange(Fw) * feat_stride, np.arange(Fh) * feat_stride)
        shifts = np.stack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()), 1).astype(f32)
        anc = (base[None, :, :] + shifts[:, None, :]).reshape(-1, 4).astype(f32)
        aw = anc[:, 2] - anc[:, 0] + f32(1)
        ah = anc[:, 3] - anc[:, 1] + f32(1)
        cx = anc[:, 0] + f32(0.5) * (aw - f32(1))
        cy = anc[:, 1] + f32(0.5) * (ah - f32(1))
        pcx = d[:, 0] * aw + cx
        pcy = d[:, 1] * ah + cy
        pw = np.exp(d[:, 2].astype(np.float64)).astype(f32) * aw  # exp in double, rounded once
        ph = np.exp(d[:, 3].astype(np.float64)).astype(f32) * ah
        boxes = np.stack((pcx - f32(0.5) * (pw - f32(1)), pcy - f32(0.5) * (ph - f32(1)),
                          pcx + f32(0.5) * (pw - f32(1)), pcy + f32(0.5) * (ph - f32(1))), 1).astype(f32)
        im_h, im_w, im_s = [f32(v) for v in im_info[b]]
        boxes[:, 0::2] = np.clip(boxes[:, 0::2], 0, im_w - 1)
        boxes[:, 1::2] = np.clip(boxes[:, 1::2], 0, im_h - 1)
        sc = fg.astype(f32).copy()
        ms = f32(min_size) * im_s
        sc[((boxes[:, 2] - boxes[:, 0] + 1) < ms) | ((boxes[:, 3] - boxes[:, 1] + 1) < ms)] = -1
        order = np.argsort(-sc, kind='stable')[:pre_nms]
        sb = np.concatenate((boxes[order], sc[order, None]), 1).astype(f32)
        keep = capi.nms_sorted(sb, nms_thresh, post_nms)
        idx = keep[np.arange(post_nms) % len(keep)]
        rois[b * post_nms:(b + 1) * post_nms, 0] = b
        rois[b * post_nms:(b + 1) * post_nms, 1:] = sb[idx, :4]
        scores_out[b * post_nms:(b + 1) * post_nms] = sb[idx, 4]
        dbg.append((order, sb, keep))
    return rois, scores_out, dbg


def proposal_targets(rois, gt_boxes, valid_ranges, post_nms, fg_thresh=0.5, stds=(0.1, 0.1, 0.2, 0.2)):
    """RoI labelling, our documented semantics: a GT (class >= 0) is *valid* for the chip
    iff lo <= sqrt(w*h) <= hi (+1 pixel convention); fg = IoU with a valid GT >= fg_thresh (label =
    class of the first arg-max GT, bbox_target = nonlinear_transform / stds, weight 1); else ignore
    (-1) if IoU with an invalid GT >= fg_thresh; else background 0."""
    R = rois.shape[0]
    f32 = np.float32
    label = np.zeros((R,), f32)
    tgt = np.zeros((R, 4), f32)
    wgt = np.zeros((R, 4), f32)
    for r in range(R):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = [f32(v) for v in rois[r, 1:]]
        area = (x2 - x1 + f32(1)) * (y2 - y1 + f32(1))
        lo, hi = [f32(v) for v in valid_ranges[b]]
        best_v, best_i, arg = f32(-1), f32(-1), -1
        for g in range(gt_boxes.shape[1]):
            gx1, gy1, gx2, gy2, c = [f32(v) for v in gt_boxes[b, g]]
            if c < 0:
                continue
            size = np.sqrt((gx2 - gx1 + f32(1)) * (gy2 - gy1 + f32(1)), dtype=f32)
            valid = size >= lo and size <= hi
            iw = min(x2, gx2) - max(x1, gx1) + f32(1)
            ov = f32(0)
            if iw > 0:
                ih = min(y2, gy2) - max(y1, gy1) + f32(1)
                if ih > 0:
                    ov = f32(iw * ih) / f32(f32(area + f32((gx2 - gx1 + f32(1)) * (gy2 - gy1 + f32(1)))) - f32(iw * ih))
            if valid:
                if ov > best_v:
                    best_v, arg = ov, g
            elif ov > best_i:
                best_i = ov
        if arg >= 0 and best_v >= f32(fg_thresh):
            gx1, gy1, gx2, gy2, c = [f32(v) for v in gt_boxes[b, arg]]
            label[r] = c
            wgt[r] = 1
            ew, eh = x2 - x1 + f32(1), y2 - y1 + f32(1)
            ecx, ecy = x1 + f32(0.5) * (ew - f32(1)), y1 + f32(0.5) * (eh - f32(1))
            gw, gh = gx2 - gx1 + f32(1), gy2 - gy1 + f32(1)
            gcx, gcy = gx1 + f32(0.5) * (gw - f32(1)), gy1 + f32(0.5) * (gh - f32(1))
            tgt[r] = [(gcx - ecx) / (ew + f32(1e-7)) / f32(stds[0]), (gcy - ecy) / (eh + f32(1e-7)) / f32(stds[1]),
                      np.log(gw / (ew + f32(1e-7))) / f32(stds[2]), np.log(gh / (eh + f32(1e-7))) / f32(stds[3])]
        elif best_i >= f32(fg_thresh):
            label[r] = -1
    return label, tgt, wgt


# ---------------------------------------------------------------------------------------------
# Mask branch (symbols/faster/resnet_mx_101_e2e_mask.py:317-318,392-395; fork operators -- spec ours, parity unpinned)
# ---------------------------------------------------------------------------------------------
def proposal_target_matches(rois, gt_boxes, valid_ranges, post, fg_thresh=0.5):
    """gt_boxes row matched by every foreground RoI (-1 otherwise): the arg-max the labelling of proposal_targets uses."""
    f32 = np.float32
    R = rois.shape[0]
    match = -np.ones((R,), np.float32)
    for r in range(R):
        b = r // post
        x1, y1, x2, y2 = [f32(v) for v in rois[r, 1:5]]
        area = (x2 - x1 + f32(1)) * (y2 - y1 + f32(1))
        lo, hi = f32(valid_ranges[b, 0]), f32(valid_ranges[b, 1])
        best_v, arg = f32(-1), -1
        for g in range(gt_boxes.shape[1]):
            gx1, gy1, gx2, gy2, c = [f32(v) for v in gt_boxes[b, g]]
            if c < 0:
                continue
            size = np.sqrt((gx2 - gx1 + f32(1)) * (gy2 - gy1 + f32(1)))
            if not (size >= lo and size <= hi):
                continue
            iw = min(x2, gx2) - max(x1, gx1) + f32(1)
            ov = f32(0)
            if iw > 0:
                ih = min(y2, gy2) - max(y1, gy1) + f32(1)
                if ih > 0:
                    ov = iw * ih / (area + (gx2 - gx1 + f32(1)) * (gy2 - gy1 + f32(1)) - iw * ih)
            if ov > best_v:
                best_v, arg = ov, g
        if arg >= 0 and best_v >= f32(fg_thresh):
            match[r] = arg
    return match


def mask_rois_select(rois, label, match, post, nm):
    """First nm foreground RoIs of every chip in RoI order, padded with [b,0,0,0,0] / -1."""
    B = rois.shape[0] // post
    mrois = np.zeros((B * nm, 5), np.float32)
    mids = -np.ones((B * nm,), np.float32)
    for b in range(B):
        mrois[b * nm:(b + 1) * nm, 0] = b
        k = 0
        for r in range(b * post, (b + 1) * post):
            if label[r] > 0 and k < nm:
                mrois[b * nm + k] = rois[r]
                mids[b * nm + k] = match[r]
                k += 1
    return mrois, mids


def mask_rcnn_target(rois, polys, ids, nm, ms=28):
    """-> targets (N, ms, ms) in {1, 0, -1}, cls (N).  Centre of RoI cell (i, j) inside the union of the matched object's
    polygons (even-odd rule per polygon, float32 crossing test as the kernel evaluates it)."""
    f32 = np.float32
    N = rois.shape[0]
    tg = -np.ones((N, ms, ms), np.float32)
    cls = np.zeros((N,), np.float32)
    for n in range(N):
        gid = int(ids[n])
        if gid < 0:
            continue
        row = polys[n // nm, gid].astype(np.float32)
        if row[0] < 0:
            continue
        cls[n] = row[0]
        nseg = int(row[1])
        if nseg <= 0:
            continue
        x1, y1 = f32(rois[n, 1]), f32(rois[n, 2])
        cw = (f32(rois[n, 3]) - x1 + f32(1)) / f32(ms)
        ch = (f32(rois[n, 4]) - y1 + f32(1)) / f32(ms)
        px = x1 + (np.arange(ms, dtype=np.float32) + f32(0.5)) * cw
        py = y1 + (np.arange(ms, dtype=np.float32) + f32(0.5)) * ch
        PX, PY = np.meshgrid(px, py)
        inside = np.zeros((ms, ms), bool)
        off = 2 + nseg
        for sgm in range(nseg):
            ln = int(row[2 + sgm])
            xs, ys = row[off:off + ln:2], row[off + 1:off + ln:2]
            nv = ln // 2
            inn = np.zeros((ms, ms), bool)
            c = nv - 1
            for a in range(nv):
                xa, ya, xc, yc = xs[a], ys[a], xs[c], ys[c]
                cross = (ya > PY) != (yc > PY)
                with np.errstate(divide='ignore', invalid='ignore'):
                    xi = (xc - xa) * (PY - ya) / (yc - ya) + xa
                inn ^= cross & (PX < xi)
                c = a
            inside |= inn
            off += ln
        tg[n] = inside.astype(np.float32)
    return tg, cls


# ---------------------------------------------------------------------------------------------
# DeformablePSROIPooling (Deformable ConvNets v1; call site :286-293 uses group_size 1; BASELINE config C4 swaps the
# head for the position-sensitive R-FCN variant, group_size = pooled_size = 7).
# data (B,C,H,W) with C = output_dim * G * G, rois (R,5), trans (R,2,P,P) or None -> out (R,output_dim,P,P).
# Bin (ph,pw) of output channel d reads data channel (d*G + gh)*G + gw, gh = floor(ph*G/P), gw = floor(pw*G/P)
# (G = 1: channel d itself).  Offsets are class-agnostic (one (2,P,P) field per RoI), part_size = pooled_size.
# ---------------------------------------------------------------------------------------------
def _ps_channels(D, G, P, ph, pw):
    gh, gw = min(max(ph * G // P, 0), G - 1), min(max(pw * G // P, 0), G - 1)
    return (np.arange(D) * G + gh) * G + gw


def _roi_bins(roi, trans, r, ph, pw, P, S, scale, trans_std):
    f32 = np.float32
    rnd = lambda v: f32(np.floor(abs(v) + 0.5) * np.sign(v))  # C round(): half away from zero
    sw, sh = rnd(roi[1]) * f32(scale) - f32(0.5), rnd(roi[2]) * f32(scale) - f32(0.5)
    ew, eh = (rnd(roi[3]) + 1) * f32(scale) - f32(0.5), (rnd(roi[4]) + 1) * f32(scale) - f32(0.5)
    rw, rh = max(ew - sw, f32(0.1)), max(eh - sh, f32(0.1))
    bw, bh = rw / P, rh / P
    tx = ty = f32(0)
    if trans is not None:
        tx, ty = trans[r, 0, ph, pw] * f32(trans_std), trans[r, 1, ph, pw] * f32(trans_std)
    return pw * bw + sw + tx * rw, ph * bh + sh + ty * rh, bw / S, bh / S, rw, rh


def dpsroi_pool(data, rois, trans, P, S, scale, trans_std=0.0, group_size=1):
    B, C, H, W = data.shape
    G = int(group_size)
    assert C % (G * G) == 0
    D = C // (G * G)
    R = rois.shape[0]
    out = np.zeros((R, D, P, P), np.float64)
    for r in range(R):
        b = int(rois[r, 0])
        for ph in range(P):
            for pw in range(P):
                ws, hs, sw_, sh_, _, _ = _roi_bins(rois[r], trans, r, ph, pw, P, S, scale, trans_std)
                acc = np.zeros(D)
                cnt = 0
                ch = _ps_channels(D, G, P, ph, pw)
                for ih in range(S):
                    for iw in range(S):
                        w, h = ws + iw * sw_, hs + ih * sh_
                        if w < -0.5 or w > W - 0.5 or h < -0.5 or h > H - 0.5:
                            continue
                        w, h = min(max(w, 0.0), W - 1.0), min(max(h, 0.0), H - 1.0)
                        x0, x1, y0, y1 = int(np.floor(w)), int(np.ceil(w)), int(np.floor(h)), int(np.ceil(h))
                        dx, dy = w - x0, h - y0
                        acc += ((1 - dx) * (1 - dy) * data[b, ch, y0, x0] + dx * (1 - dy) * data[b, ch, y0, x1] +
                                (1 - dx) * dy * data[b, ch, y1, x0] + dx * dy * data[b, ch, y1, x1])
                        cnt += 1
                if cnt:
                    out[r, :, ph, pw] = acc / cnt
    return out


def dpsroi_pool_backward(dout, data, rois, trans, P, S, scale, trans_std=0.0, group_size=1):
    B, C, H, W = data.shape
    G = int(group_size)
    D = C // (G * G)
    R = rois.shape[0]
    d_data = np.zeros(data.shape, np.float64)
    d_trans = None if trans is None else np.zeros(trans.shape, np.float64)
    for r in range(R):
        b = int(rois[r, 0])
        for ph in range(P):
            for pw in range(P):
                ws, hs, sw_, sh_, rw, rh = _roi_bins(rois[r], trans, r, ph, pw, P, S, scale, trans_std)
                pts = []
                for ih in range(S):
                    for iw in range(S):
                        w, h = ws + iw * sw_, hs + ih * sh_
                        if w < -0.5 or w > W - 0.5 or h < -0.5 or h > H - 0.5:
                            continue
                        pts.append((min(max(w, 0.0), W - 1.0), min(max(h, 0.0), H - 1.0)))
                if not pts:
                    continue
                dv = dout[r, :, ph, pw] / len(pts)
                ch = _ps_channels(D, G, P, ph, pw)
                for w, h in pts:
                    x0, x1, y0, y1 = int(np.floor(w)), int(np.ceil(w)), int(np.floor(h)), int(np.ceil(h))
                    dx, dy = w - x0, h - y0
                    d_data[b, ch, y0, x0] += (1 - dx) * (1 - dy) * dv
                    d_data[b, ch, y0, x1] += dx * (1 - dy) * dv
                    d_data[b, ch, y1, x0] += (1 - dx) * dy * dv
                    d_data[b, ch, y1, x1] += dx * dy * dv
                    if trans is not None:
                        U00, U01, U10, U11 = data[b, ch, y0, x0], data[b, ch, y0, x1], data[b, ch, y1, x0], data[b, ch, y1, x1]
                        d_trans[r, 0, ph, pw] += np.sum((U11 * dy + U01 * (1 - dy) - U10 * dy - U00 * (1 - dy)) * dv) * trans_std * rw
                        d_trans[r, 1, ph, pw] += np.sum((U11 * dx + U10 * (1 - dx) - U01 * dx - U00 * (1 - dx)) * dv) * trans_std * rh
    return d_data, d_trans


# ---- the same operator as three sparse sampling matrices (for BASELINE sizes: R = 6000 RoIs take minutes in the loops above).
# Row (r, ph, pw), column (b, y, x):  A = bilinear weights / valid-sample count,  Gx / Gy = d(sample)/dx, /dy weights / count.
# forward  out[r, :, ph, pw] = A   @ data[(b, y, x), channels of the bin]
# backward d_data           += A.T @ dout;   d_trans[r, 0|1, ph, pw] = <Gx|Gy @ data, dout> * trans_std * roi_w|roi_h
# Sample coordinates are computed in float32 exactly as _roi_bins / dpsroi_pool do; the products are then carried in float64
# (the loops multiply float32 weights into float32 data first), so the two agree to ~1e-6, not to the bit:
# tests/test_oracle_graph_cpu.py pins this restatement to the loop definition.
def _dpsroi_operators(rois, trans, B, H, W, P, S, scale, trans_std):
    import scipy.sparse as sp
    f32 = np.float32
    R = rois.shape[0]
    rois = np.asarray(rois, np.float32)
    rnd = lambda v: (np.floor(np.abs(v) + f32(0.5)) * np.sign(v)).astype(f32)
    sw, sh = rnd(rois[:, 1]) * f32(scale) - f32(0.5), rnd(rois[:, 2]) * f32(scale) - f32(0.5)
    ew, eh = (rnd(rois[:, 3]) + f32(1)) * f32(scale) - f32(0.5), (rnd(rois[:, 4]) + f32(1)) * f32(scale) - f32(0.5)
    rw, rh = np.maximum(ew - sw, f32(0.1)), np.maximum(eh - sh, f32(0.1))
    bw, bh = (rw / f32(P)).astype(f32), (rh / f32(P)).astype(f32)
    idx = np.arange(P, dtype=f32)
    if trans is not None:
        tx = (np.asarray(trans, f32)[:, 0] * f32(trans_std)).astype(f32)
        ty = (np.asarray(trans, f32)[:, 1] * f32(trans_std)).astype(f32)
    else:
        tx = ty = np.zeros((R, P, P), f32)
    ws = (idx[None, None, :] * bw[:, None, None] + sw[:, None, None] + tx * rw[:, None, None]).astype(f32)      # (R, ph, pw)
    hs = (idx[None, :, None] * bh[:, None, None] + sh[:, None, None] + ty * rh[:, None, None]).astype(f32)
    ssw, ssh = (bw / f32(S)).astype(f32), (bh / f32(S)).astype(f32)
    b = rois[:, 0].astype(np.int64)
    rows = np.arange(R * P * P).reshape(R, P, P)
    cnt = np.zeros((R, P, P), np.int64)
    rr, cc, va, vx, vy = [], [], [], [], []
    for ih in range(S):
        for iw in range(S):
            w = (ws + f32(iw) * ssw[:, None, None]).astype(f32)
            h = (hs + f32(ih) * ssh[:, None, None]).astype(f32)
            ok = ~((w < -0.5) | (w > W - 0.5) | (h < -0.5) | (h > H - 0.5))
            w = np.minimum(np.maximum(w, f32(0.0)), f32(W - 1.0))
            h = np.minimum(np.maximum(h, f32(0.0)), f32(H - 1.0))
            x0, x1, y0, y1 = np.floor(w).astype(np.int64), np.ceil(w).astype(np.int64), np.floor(h).astype(np.int64), np.ceil(h).astype(np.int64)
            dx, dy = (w - x0).astype(f32).astype(np.float64), (h - y0).astype(f32).astype(np.float64)
            cnt += ok
            base = b[:, None, None] * H
            for (yy, xx, wa, wx, wy) in ((y0, x0, (1 - dx) * (1 - dy), -(1 - dy), -(1 - dx)), (y0, x1, dx * (1 - dy), (1 - dy), -dx),
                                         (y1, x0, (1 - dx) * dy, -dy, (1 - dx)), (y1, x1, dx * dy, dy, dx)):
                rr.append(rows[ok]); cc.append(((base + yy) * W + xx)[ok]); va.append(wa[ok]); vx.append(wx[ok]); vy.append(wy[ok])
    rr, cc = np.concatenate(rr), np.concatenate(cc)
    inv = 1.0 / np.maximum(cnt, 1).reshape(-1)
    shape = (R * P * P, B * H * W)
    mk = lambda v: sp.diags(inv) @ sp.csr_matrix((np.concatenate(v), (rr, cc)), shape=shape)
    return mk(va), mk(vx), mk(vy), rw.astype(np.float64), rh.astype(np.float64)


def _ps_groups(G, P):
    """bins (ph, pw) by the map group (gh, gw) they read (position-sensitive pooling); one group holding all bins for G = 1"""
    groups = {}
    for ph in range(P):
        for pw in range(P):
            gh, gw = min(max(ph * G // P, 0), G - 1), min(max(pw * G // P, 0), G - 1)
            groups.setdefault((gh, gw), []).append(ph * P + pw)
    return groups


def dpsroi_pool_fast(data, rois, trans, P, S, scale, trans_std=0.0, group_size=1):
    B, C, H, W = data.shape
    G = int(group_size)
    D, R = C // (G * G), rois.shape[0]
    A, _, _, _, _ = _dpsroi_operators(rois, trans, B, H, W, P, S, scale, trans_std)
    flat = np.ascontiguousarray(np.asarray(data, np.float64).transpose(0, 2, 3, 1)).reshape(B * H * W, C)
    out = np.zeros((R, P * P, D))
    for (gh, gw), bins in _ps_groups(G, P).items():
        ch = (np.arange(D) * G + gh) * G + gw
        rsel = (np.arange(R)[:, None] * (P * P) + np.asarray(bins)[None, :]).reshape(-1)
        out[:, bins, :] = (A[rsel] @ flat[:, ch]).reshape(R, len(bins), D)
    return out.transpose(0, 2, 1).reshape(R, D, P, P)


def dpsroi_pool_backward_fast(dout, data, rois, trans, P, S, scale, trans_std=0.0, group_size=1):
    B, C, H, W = data.shape
    G = int(group_size)
    D, R = C // (G * G), rois.shape[0]
    A, GX, GY, rw, rh = _dpsroi_operators(rois, trans, B, H, W, P, S, scale, trans_std)
    flat = np.ascontiguousarray(np.asarray(data, np.float64).transpose(0, 2, 3, 1)).reshape(B * H * W, C)
    dflat = np.zeros((B * H * W, C))
    do = np.asarray(dout, np.float64).reshape(R, D, P * P).transpose(0, 2, 1)          # (R, bins, D)
    d_trans = None if trans is None else np.zeros((R, 2, P * P))
    for (gh, gw), bins in _ps_groups(G, P).items():
        ch = (np.arange(D) * G + gh) * G + gw
        rsel = (np.arange(R)[:, None] * (P * P) + np.asarray(bins)[None, :]).reshape(-1)
        dv = np.ascontiguousarray(do[:, bins, :]).reshape(-1, D)
        dflat[:, ch] += A[rsel].T @ dv
        if trans is not None:
            sub = flat[:, ch]
            d_trans[:, 0, bins] = ((GX[rsel] @ sub) * dv).sum(1).reshape(R, len(bins)) * trans_std * rw[:, None]
            d_trans[:, 1, bins] = ((GY[rsel] @ sub) * dv).sum(1).reshape(R, len(bins)) * trans_std * rh[:, None]
    d_data = dflat.reshape(B, H, W, C).transpose(0, 3, 1, 2)
    return d_data, (None if trans is None else d_trans.reshape(R, 2, P, P))


# ---------------------------------------------------------------------------------------------
# DeformableConvolution v1 sampling (call site :124-128): column tensor (N, Ho, Wo, T, C)
# ---------------------------------------------------------------------------------------------
def _deform_sample(py, px, H, W):
    ok = py >= 0 and px >= 0 and py < H and px < W
    y0, x0 = int(np.floor(py)), int(np.floor(px))
    if y0 >= H - 1:
        y0 = y1 = H - 1
        ly = 0.0
    else:
        y1, ly = y0 + 1, py - y0
    if x0 >= W - 1:
        x0 = x1 = W - 1
        lx = 0.0
    else:
        x1, lx = x0 + 1, px - x0
    return ok, y0, y1, x0, x1, ly, lx


def _deform_positions(offset, n, g, T, KW, Ho, Wo, stride, pad, dil, H, W):
    """Sampling geometry of image n, deformable group g, all (tap, oy, ox) at once -- the branches of _deform_sample on arrays.
    -> ok (T,Ho,Wo) bool, y0, y1, x0, x1 (int, in range everywhere), ly, lx (float64)."""
    t = np.arange(T)
    kh, kw = t // KW, t % KW
    by = (np.arange(Ho)[None, :] * stride - pad + kh[:, None] * dil).astype(np.float32)          # (T, Ho)
    bx = (np.arange(Wo)[None, :] * stride - pad + kw[:, None] * dil).astype(np.float32)          # (T, Wo)
    off = np.asarray(offset[n, g * 2 * T:(g + 1) * 2 * T]).reshape(T, 2, Ho, Wo)
    # the sampling position is a float32 sum (the operator's DType; the border test `p < dim` is a discontinuity: 32 - 1e-6 is
    # 32.0 in float32 and inside the map in double)
    py = (by[:, :, None] + off[:, 0].astype(np.float32)).astype(np.float64)
    px = (bx[:, None, :] + off[:, 1].astype(np.float32)).astype(np.float64)
    ok = (py >= 0) & (px >= 0) & (py < H) & (px < W)
    y0 = np.floor(py).astype(np.int64)
    x0 = np.floor(px).astype(np.int64)
    top, right = y0 >= H - 1, x0 >= W - 1
    ly = np.where(top, 0.0, py - y0)
    lx = np.where(right, 0.0, px - x0)
    y0 = np.where(top, H - 1, y0)
    x0 = np.where(right, W - 1, x0)
    y1 = np.where(top, H - 1, y0 + 1)
    x1 = np.where(right, W - 1, x0 + 1)
    clip = lambda a, hi: np.clip(a, 0, hi)                     # (positions outside the map are masked by `ok`; keep them indexable)
    return ok, clip(y0, H - 1), clip(y1, H - 1), clip(x0, W - 1), clip(x1, W - 1), ly, lx


def deform_im2col(data, offset, KH, KW, stride, pad, dil, DG):
    """data (N,C,H,W), offset (N, 2*T*DG, Ho, Wo) -> col (N,Ho,Wo,T,C).  Array form of deform_im2col_loops (the statement the
    array form is checked against in tests/test_oracle_golden.py): same positions, same four-corner weights."""
    N, C, H, W = data.shape
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    T, cg = KH * KW, C // DG
    col = np.zeros((N, Ho, Wo, T, C), np.float64)
    for n in range(N):
        for g in range(DG):
            ok, y0, y1, x0, x1, ly, lx = _deform_positions(offset, n, g, T, KW, Ho, Wo, stride, pad, dil, H, W)
            d = np.asarray(data[n, g * cg:(g + 1) * cg], np.float64).transpose(1, 2, 0)          # (H, W, cg)
            w = [(1 - ly) * (1 - lx), (1 - ly) * lx, ly * (1 - lx), ly * lx]
            v = ((w[0] * ok)[..., None] * d[y0, x0] + (w[1] * ok)[..., None] * d[y0, x1] +
                 (w[2] * ok)[..., None] * d[y1, x0] + (w[3] * ok)[..., None] * d[y1, x1])         # (T, Ho, Wo, cg)
            col[n, :, :, :, g * cg:(g + 1) * cg] = v.transpose(1, 2, 0, 3)
    return col


def deform_col2im(dcol, data, offset, KH, KW, stride, pad, dil, DG):
    """Gradients of deform_im2col w.r.t. data and offset; array form of deform_col2im_loops (the four-corner scatter as a sparse
    (pixels x samples) product, float64: the sums differ from the loop's order of addition by rounding only)."""
    import scipy.sparse as sp
    N, C, H, W = data.shape
    _, Ho, Wo, T, _ = dcol.shape
    cg = C // DG
    d_data = np.zeros(data.shape, np.float64)
    d_off = np.zeros(offset.shape, np.float64)
    S = T * Ho * Wo
    for n in range(N):
        for g in range(DG):
            ok, y0, y1, x0, x1, ly, lx = _deform_positions(offset, n, g, T, KW, Ho, Wo, stride, pad, dil, H, W)
            cs = slice(g * cg, (g + 1) * cg)
            dd = np.asarray(dcol[n, :, :, :, cs], np.float64).transpose(2, 0, 1, 3).reshape(S, cg)     # samples (t, oy, ox) x channels
            okf = ok.reshape(S).astype(np.float64)
            w = [((1 - ly) * (1 - lx)).reshape(S) * okf, ((1 - ly) * lx).reshape(S) * okf, (ly * (1 - lx)).reshape(S) * okf,
                 (ly * lx).reshape(S) * okf]
            pix = [(y0 * W + x0).reshape(S), (y0 * W + x1).reshape(S), (y1 * W + x0).reshape(S), (y1 * W + x1).reshape(S)]
            M = sp.coo_matrix((np.concatenate(w), (np.concatenate(pix), np.tile(np.arange(S), 4))), shape=(H * W, S)).tocsr()
            d_data[n, cs] = (M @ dd).reshape(H, W, cg).transpose(2, 0, 1)
            d = np.asarray(data[n, cs], np.float64).transpose(1, 2, 0)
            a, b, c, e = d[y0, x0].reshape(S, cg), d[y0, x1].reshape(S, cg), d[y1, x0].reshape(S, cg), d[y1, x1].reshape(S, cg)
            lyf, lxf = ly.reshape(S, 1), lx.reshape(S, 1)
            gy = (dd * ((1 - lxf) * (c - a) + lxf * (e - b))).sum(1) * okf
            gx = (dd * ((1 - lyf) * (b - a) + lyf * (e - c))).sum(1) * okf
            d_off[n, g * 2 * T:(g + 1) * 2 * T] = np.stack((gy.reshape(T, Ho, Wo), gx.reshape(T, Ho, Wo)), 1).reshape(2 * T, Ho, Wo)
    return d_data, d_off


def deform_im2col_loops(data, offset, KH, KW, stride, pad, dil, DG):
    """data (N,C,H,W), offset (N, 2*T*DG, Ho, Wo) -> col (N,Ho,Wo,T,C)."""
    N, C, H, W = data.shape
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    T, cg = KH * KW, C // DG
    col = np.zeros((N, Ho, Wo, T, C), np.float64)
    for n in range(N):
        for oy in range(Ho):
            for ox in range(Wo):
                for t in range(T):
                    kh, kw = divmod(t, KW)
                    for g in range(DG):
                        # the sampling position is a float32 sum (the operator's DType; the border test `p < dim` is a
                        # discontinuity: 32 - 1e-6 is 32.0 in float32 and inside the map in double)
                        py = float(np.float32(oy * stride - pad + kh * dil) + np.float32(offset[n, g * 2 * T + 2 * t, oy, ox]))
                        px = float(np.float32(ox * stride - pad + kw * dil) + np.float32(offset[n, g * 2 * T + 2 * t + 1, oy, ox]))
                        ok, y0, y1, x0, x1, ly, lx = _deform_sample(py, px, H, W)
                        if not ok:
                            continue
                        cs = slice(g * cg, (g + 1) * cg)
                        col[n, oy, ox, t, cs] = ((1 - ly) * (1 - lx) * data[n, cs, y0, x0] + (1 - ly) * lx * data[n, cs, y0, x1] +
                                                 ly * (1 - lx) * data[n, cs, y1, x0] + ly * lx * data[n, cs, y1, x1])
    return col


def deform_col2im_loops(dcol, data, offset, KH, KW, stride, pad, dil, DG):
    N, C, H, W = data.shape
    _, Ho, Wo, T, _ = dcol.shape
    cg = C // DG
    d_data = np.zeros(data.shape, np.float64)
    d_off = np.zeros(offset.shape, np.float64)
    for n in range(N):
        for oy in range(Ho):
            for ox in range(Wo):
                for t in range(T):
                    kh, kw = divmod(t, KW)
                    for g in range(DG):
                        # the sampling position is a float32 sum (the operator's DType; the border test `p < dim` is a
                        # discontinuity: 32 - 1e-6 is 32.0 in float32 and inside the map in double)
                        py = float(np.float32(oy * stride - pad + kh * dil) + np.float32(offset[n, g * 2 * T + 2 * t, oy, ox]))
                        px = float(np.float32(ox * stride - pad + kw * dil) + np.float32(offset[n, g * 2 * T + 2 * t + 1, oy, ox]))
                        ok, y0, y1, x0, x1, ly, lx = _deform_sample(py, px, H, W)
                        if not ok:
                            continue
                        cs = slice(g * cg, (g + 1) * cg)
                        d = dcol[n, oy, ox, t, cs]
                        d_data[n, cs, y0, x0] += (1 - ly) * (1 - lx) * d
                        d_data[n, cs, y0, x1] += (1 - ly) * lx * d
                        d_data[n, cs, y1, x0] += ly * (1 - lx) * d
                        d_data[n, cs, y1, x1] += ly * lx * d
                        a, b, c, e = data[n, cs, y0, x0], data[n, cs, y0, x1], data[n, cs, y1, x0], data[n, cs, y1, x1]
                        d_off[n, g * 2 * T + 2 * t, oy, ox] = np.sum(d * ((1 - lx) * (c - a) + lx * (e - b)))
                        d_off[n, g * 2 * T + 2 * t + 1, oy, ox] = np.sum(d * ((1 - ly) * (b - a) + ly * (e - c)))
    return d_data, d_off
