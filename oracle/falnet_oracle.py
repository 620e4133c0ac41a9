"""CPU oracle for the FAL_netB hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A PyTorch-CPU fp32 restatement of the reference's algorithm for the path named by
BASELINE.json's north_star: FAL_netB forward, the MED head
(softmax over N planes, plane-sweep warp, blend, occlusion masks), the
L1 + VGG-perceptual + edge-aware-smoothness losses, the Stage-1 / Stage-2 step
bodies, `ms_pp`, and the KITTI metric chain.  Every function cites the reference
file:line it follows (paths relative to /root/reference).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker / the reported CPU baseline.  The
product (`fal_net_amd/`) never imports it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors, so
this oracle is pinned against outputs of the reference itself, produced in the
build container by importing /root/reference (`tests/golden/make_goldens.py`)
and committed as `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks them.
Third-party arithmetic: torch (conv2d/softmax/Adam semantics; torch 2.10 is the
oracle for those) and the torchvision VGG19 cfg-"E" architecture, restated here;
perceptual-loss parity with the real ImageNet weights is *parity unpinned* (weights
are neither vendored nor downloadable) -- pinned for seeded VGG weights only.

Arithmetic is fp32 throughout, like the reference (no autocast anywhere in it).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- backbone
def _elu(x):
    return F.elu(x)


def backbone_prefix(sd):
    """Attribute name of the backbone in the checkpoint keys: `backbone` (FAL_netB.py:184), `BackBone` (FAL_netA.py:183)
    or `synth` (FAL_netC.py:185).  The three variants share BackBone.forward; channel counts and the residual-conv
    kernel shapes (3x1 / 1x3 in FAL_netA.py:73-75) are read off the weights themselves."""
    for k in sd:
        if k.endswith(".conv0.0.weight"):
            return k[:-len(".conv0.0.weight")]
    raise KeyError("no <backbone>.conv0.0.weight in the state dict")


def _conv(sd, key, x, stride=1):
    """'same'-padded conv (3x3 everywhere; 3x1 / 1x3 in FAL_netA's residual blocks), bias if present in the checkpoint
    (FAL_netB.py:44-47, FAL_netA.py:73-76)."""
    w = sd[key + ".weight"]
    return F.conv2d(x, w, sd.get(key + ".bias"), stride=stride, padding=(w.shape[2] // 2, w.shape[3] // 2))


def _conv_elu(sd, name, x, stride):
    # conv_elu, batchNorm=False branch: FAL_netB.py:44-48
    return _elu(_conv(sd, f"{name}.0", x, stride))


def _res_block(sd, name, x):
    # residual_block.forward: FAL_netB.py:78-80
    h = _elu(_conv(sd, f"{name}.conv1", x))
    return _elu(_conv(sd, f"{name}.conv2", h) + x)


def _deconv(sd, name, x, ref):
    # deconv.forward: nearest resize to the skip's size then conv+ELU, FAL_netB.py:57-60
    x = F.interpolate(x, size=(ref.size(2), ref.size(3)), mode="nearest")
    return _elu(_conv(sd, f"{name}.conv1", x))


class _Prefixed:
    """state-dict view that prepends the backbone prefix to every key."""

    def __init__(self, sd, prefix):
        self.sd, self.p = sd, prefix + "."

    def __getitem__(self, k):
        return self.sd[self.p + k]

    def get(self, k, default=None):
        return self.sd.get(self.p + k, default)


def backbone_forward(sd, x, flow):
    """BackBone.forward, FAL_netB.py:140-176 (= FAL_netA.py:139-175, FAL_netC.py:141-177) -> raw logits `dlog` (B,N,H,W)."""
    sd = _Prefixed(sd, backbone_prefix(sd))
    c0 = _res_block(sd, "conv0_1", _conv_elu(sd, "conv0", x, 1))
    c1 = _res_block(sd, "conv1_1", _conv_elu(sd, "conv1", torch.cat((c0, flow), 1), 2))
    c2 = _res_block(sd, "conv2_1", _conv_elu(sd, "conv2", c1, 2))
    c3 = _res_block(sd, "conv3_1", _conv_elu(sd, "conv3", c2, 2))
    c4 = _res_block(sd, "conv4_1", _conv_elu(sd, "conv4", c3, 2))
    c5 = _res_block(sd, "conv5_1", _conv_elu(sd, "conv5", c4, 2))
    c6 = _res_block(sd, "conv6_1", _conv_elu(sd, "conv6", c5, 2))
    i6 = _conv_elu(sd, "iconv6", torch.cat((_deconv(sd, "deconv6", c6, c5), c5), 1), 1)
    i5 = _conv_elu(sd, "iconv5", torch.cat((_deconv(sd, "deconv5", i6, c4), c4), 1), 1)
    i4 = _conv_elu(sd, "iconv4", torch.cat((_deconv(sd, "deconv4", i5, c3), c3), 1), 1)
    i3 = _conv_elu(sd, "iconv3", torch.cat((_deconv(sd, "deconv3", i4, c2), c2), 1), 1)
    i2 = _conv_elu(sd, "iconv2", torch.cat((_deconv(sd, "deconv2", i3, c1), c1), 1), 1)
    cat1 = torch.cat((_deconv(sd, "deconv1", i2, c0), c0), 1)
    return F.conv2d(cat1, sd["iconv1.weight"], None, padding=1)  # FAL_netB.py:127,174


# --------------------------------------------------------------------------- MED head
def plane_disparities(min_disp, max_disp, n_planes):
    """d_n = max*exp(ln(max/min)*(n/(N-1)-1)), per sample (FAL_netB.py:223-225). -> (B,N)."""
    c = torch.arange(n_planes, dtype=torch.float32) / (n_planes - 1)
    mx, mn = max_disp.reshape(-1, 1), min_disp.reshape(-1, 1)
    return mx * torch.exp(torch.log(mx / mn) * (c.view(1, -1) - 1.0))


def shift_planes(t, s):
    """Horizontal 2-tap shift with zero padding: the closed form of the reference's
    affine_grid + grid_sample(bilinear, zeros, align_corners=True) with a pure x offset
    (FAL_netB.py:231-247).

    out[b,c,y,x] = (1-a)*t[b,c,y,x+k] + a*t[b,c,y,x+k+1], k=floor(s), a=s-k, zero outside
    [0,W-1].  `t` (B,C,H,W), `s` (B,C) in pixels (may be negative).
    """
    B, C, H, W = t.shape
    k = torch.floor(s)
    a = (s - k).view(B, C, 1, 1)
    k = k.long().view(B, C, 1)
    x = torch.arange(W).view(1, 1, W)
    i0 = x + k
    i1 = i0 + 1
    tp = F.pad(t, (1, 1))  # index -1 and W map to the zero columns

    def tap(idx):
        oob = (idx < 0) | (idx > W - 1)
        idx = torch.where(oob, torch.full_like(idx, -1), idx) + 1
        return torch.gather(tp, 3, idx.view(B, C, 1, W).expand(B, C, H, W))

    return (1.0 - a) * tap(i0) + a * tap(i1)


def _maskr_align_corners_false(sm, d):
    """FAL_netA.py:264 samples softmax(dlog0) for maskR with grid_sample's DEFAULT align_corners=False on a grid built
    with align_corners=True (:231,:241-242): pixel (x, y) reads (x*W/(W-1) + d_n - 0.5, y*H/(H-1) - 0.5), bilinear in both
    axes with zero padding.  Restated with torch's own grid_sample (torch is the oracle for that arithmetic)."""
    B, N, H, W = sm.shape
    th = torch.zeros(B, 2, 3)
    th[:, 0, 0] = 1
    th[:, 1, 1] = 1
    grid = F.affine_grid(th, [B, 1, H, W], align_corners=True)
    acc = 0
    for n in range(N):
        g = grid.clone()
        g[:, :, :, 0] = g[:, :, :, 0] + (2 * d[:, n] / W).view(B, 1, 1)
        acc = acc + F.grid_sample(sm[:, n:n + 1], g, align_corners=False)
    return acc


def med_head(dlog0, left, min_disp, max_disp, ret_disp=True, ret_subocc=False, ret_pan=False, maskr_align_corners=True):
    """FAL_net.forward after conv0 (FAL_netB.py:216-297) in closed form.

    Returns a dict with the requested tensors: disp, p_im0, maskL, maskR, Dprob.
    """
    B, N, H, W = dlog0.shape
    out = {}
    d = plane_disparities(min_disp, max_disp, N)  # (B,N) pixels
    sm = torch.softmax(dlog0, 1)  # :216
    if ret_disp:
        out["disp"] = (d.view(B, N, 1, 1) * sm).sum(1, keepdim=True)  # :219-226
    if not (ret_subocc or ret_pan):
        return out
    # x_of = 2 d/W normalised; align_corners=True un-normalises by (W-1)/2 -> d (W-1)/W px
    s = d * (W - 1) / W
    dprob = torch.softmax(shift_planes(dlog0, s), 1)  # :236-248 (OOB logit is 0, not -inf)
    out["Dprob"] = dprob
    if ret_pan:  # :280-282
        p = 0
        for n in range(N):
            p = p + shift_planes(left, s[:, n:n + 1].expand(B, left.shape[1])) * dprob[:, n:n + 1]
        out["p_im0"] = p
    if ret_subocc:  # :264-273,291-292 (no grad)
        with torch.no_grad():
            if maskr_align_corners:
                out["maskR"] = shift_planes(sm.detach(), s).sum(1, keepdim=True).clamp(max=1.0)
            else:
                out["maskR"] = _maskr_align_corners_false(sm.detach(), d).clamp(max=1.0)
            out["maskL"] = shift_planes(dprob.detach(), -s).sum(1, keepdim=True).clamp(max=1.0)
    return out


def falnet_forward(sd, left, min_disp, max_disp, ret_disp=True, ret_subocc=False, ret_pan=False,
                   return_dict=False):
    """FAL_net.forward (FAL_netB.py:200-297). Same return convention as the reference:
    bare `disp` when only ret_disp, else list [p_im0][disp][maskL, maskR]."""
    B, C, H, W = left.shape
    flow = (max_disp.view(B, 1, 1, 1) / 100.0).expand(B, 1, H, W)  # :208-209
    dlog = backbone_forward(sd, left, flow)
    dlog0 = F.conv2d(dlog, sd["conv0.weight"], sd["conv0.bias"])  # :215
    out = med_head(dlog0, left, min_disp, max_disp, ret_disp, ret_subocc, ret_pan,
                   maskr_align_corners=backbone_prefix(sd) != "BackBone")  # FAL_netA.py:264
    out["dlog0"] = dlog0
    if return_dict:
        return out
    if ret_disp and not ret_subocc and not ret_pan:
        return out["disp"]
    res = []
    if ret_pan:
        res.append(out["p_im0"])
    if ret_disp:
        res.append(out["disp"])
    if ret_subocc:
        res += [out["maskL"], out["maskR"]]
    return res


# --------------------------------------------------------------------------- losses
_VGG_SLICES = ((0, 2), (5, 7), (10, 12, 14, 16))  # conv indices per slice; each slice ends in MaxPool


def vgg_forward(vsd, x):
    """Vgg19_pc.forward (loss_functions.py:36-44): torchvision cfg-E features[0:5],[5:10],[10:19].
    Each slice = (conv3x3+ReLU)* then MaxPool2d(2) (loss_functions.py:21-29)."""
    outs = []
    for convs in _VGG_SLICES:
        for i in convs:
            x = F.relu(F.conv2d(x, vsd[f"features.{i}.weight"], vsd[f"features.{i}.bias"], padding=1))
        x = F.max_pool2d(x, 2)
        outs.append(x)
    return tuple(outs)


def perceptual_loss(out_vgg, label_vgg):
    # loss_functions.py:59-67 (layer=None branch)
    return sum(torch.mean((out_vgg[i] - label_vgg[i]) ** 2) for i in range(3))


def rec_loss_fnc(vsd, mask, synth, label, vgg_label, a_p):
    # loss_functions.py:52-56
    loss = torch.mean(mask * torch.abs(synth - label))
    if a_p > 0 and vgg_label is not None:
        loss = loss + a_p * perceptual_loss(vgg_forward(vsd, mask * synth + (1 - mask) * label), vgg_label)
    return loss


def smoothness(img, disp, gamma=1):
    """Edge-aware smoothness, loss_functions.py:70-101 (+ getGrayscale :104-109).
    Zero padding 1 is applied to the already-cropped tensors (conv2d padding=1)."""
    mean = torch.tensor([0.411, 0.432, 0.45]).view(1, 3, 1, 1)
    rgb = img + mean
    g = (0.299 * rgb[:, 0] + 0.587 * rgb[:, 1] + 0.114 * rgb[:, 2]).unsqueeze(1).detach()
    gp = F.pad(g, (1, 1, 1, 1))
    dp = F.pad(disp, (1, 1, 1, 1))
    c = (slice(None), slice(None), slice(1, -1), slice(1, -1))
    dx_img = -gp[:, :, 1:-1, :-2] + 2 * gp[c] - gp[:, :, 1:-1, 2:]   # sx_filter [-1,2,-1]
    dy_img = -gp[:, :, :-2, 1:-1] + 2 * gp[c] - gp[:, :, 2:, 1:-1]   # sy_filter
    dx_d = dp[c] - dp[:, :, 1:-1, 2:]    # dx_filter  [0,1,-1]
    dx1_d = dp[c] - dp[:, :, 1:-1, :-2]  # dx1_filter [-1,1,0]
    dy_d = dp[c] - dp[:, :, :-2, 1:-1]   # dy_filter  (-1 above)
    dy1_d = dp[c] - dp[:, :, 2:, 1:-1]   # dy1_filter (-1 below)
    return torch.mean((dx_d.abs() + dx1_d.abs()) * torch.exp(-gamma * dx_img.abs()) +
                      (dy_d.abs() + dy1_d.abs()) * torch.exp(-gamma * dy_img.abs()))


# --------------------------------------------------------------------------- step bodies
class OracleAdam:
    """torch.optim.Adam over the reference's two param groups (Train_Stage1_K.py:177-180)."""

    def __init__(self, params, lr=1e-4, betas=(0.5, 0.999)):
        names = list(params.keys())
        groups = [{"params": [params[k] for k in names if "bias" in k], "weight_decay": 0.0},
                  {"params": [params[k] for k in names if "weight" in k], "weight_decay": 0.0}]
        self.opt = torch.optim.Adam(groups, lr=lr, betas=betas)

    def step(self):
        self.opt.step()

    def zero_grad(self):
        self.opt.zero_grad()


def leaf_params(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def stage1_losses(params, vsd, left, right, min_disp, max_disp, a_p=0.01, a_sm=0.2 * 2 / 512):
    """Loss part of one Stage-1 iteration, Train_Stage1_K.py:236-258."""
    W = left.shape[3]
    rpan, ldisp = falnet_forward(params, left, min_disp, max_disp, ret_disp=True, ret_pan=True)
    with torch.no_grad():
        vgg_right = vgg_forward(vsd, right) if a_p > 0 else None  # :241-244
    rec = rec_loss_fnc(vsd, 1, rpan, right, vgg_right, a_p)  # :247-248
    c = int(0.20 * W)
    sm = smoothness(left[:, :, :, c:], ldisp[:, :, :, c:], gamma=2) if a_sm > 0 else 0  # :252-255
    loss = rec + a_sm * sm  # :258
    return {"loss": loss, "rec": rec, "sm": sm, "rpan": rpan, "ldisp": ldisp}


def stage1_step(params, opt, vsd, left, right, min_disp, max_disp, a_p=0.01, a_sm=0.2 * 2 / 512):
    """zero_grad -> forward -> losses -> backward -> Adam (Train_Stage1_K.py:233-262)."""
    opt.zero_grad()
    out = stage1_losses(params, vsd, left, right, min_disp, max_disp, a_p, a_sm)
    out["loss"].backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in params.items()}
    opt.step()
    out["grads"] = grads
    return out


def hflip(x):
    """The reference flips with affine_grid+grid_sample (x negated); that equals an index
    reversal to <=8.4e-7 abs (Train_Stage2_K.py:248-253)."""
    return torch.flip(x, [3])


def stage1_slow_losses(params, vsd, left, right, min_disp, max_disp, a_p=0.01, a_sm=0.2 * 2 / 512):
    """Loss part of one iteration of the two-view Stage-1 variant, Train_Stage1_Kslow.py:243-281:
    the batch is (left | flip(right)); the second half of every output is flipped back."""
    B, C, H, W = left.shape
    mn2, mx2 = torch.cat((min_disp, min_disp), 0), torch.cat((max_disp, max_disp), 0)
    pan, disp = falnet_forward(params, torch.cat((left, hflip(right)), 0), mn2, mx2,
                               ret_disp=True, ret_pan=True)  # :245-248
    rpan, lpan = pan[0:B], hflip(pan[B:])  # :249-256
    ldisp, rdisp = disp[0:B], hflip(disp[B:])
    with torch.no_grad():
        vgg_right, vgg_left = (vgg_forward(vsd, right), vgg_forward(vsd, left)) if a_p > 0 else (None, None)  # :259-264
    rec = (rec_loss_fnc(vsd, 1, rpan, right, vgg_right, a_p) +
           rec_loss_fnc(vsd, 1, lpan, left, vgg_left, a_p)) / 2  # :268-269
    c2, c8 = int(0.20 * W), int(0.80 * W)
    sm = 0
    if a_sm > 0:  # :274-278
        sm = (smoothness(left[:, :, :, c2:], ldisp[:, :, :, c2:], gamma=2) +
              smoothness(right[:, :, :, 0:c8], rdisp[:, :, :, 0:c8], gamma=2)) / 2
    loss = rec + a_sm * sm  # :281
    return {"loss": loss, "rec": rec, "sm": sm, "rpan": rpan, "lpan": lpan, "ldisp": ldisp, "rdisp": rdisp}


def stage2_losses(params, teacher_sd, vsd, left, right, min_disp, max_disp,
                  a_p=0.01, a_sm=0.4 * 2 / 512, a_mr=1.0):
    """Loss part of one Stage-2 iteration, Train_Stage2_K.py:247-329."""
    B, C, H, W = left.shape
    mn2, mx2 = torch.cat((min_disp, min_disp), 0), torch.cat((max_disp, max_disp), 0)
    with torch.no_grad():  # :256-264 frozen teacher, disparity only
        tdisp = falnet_forward(teacher_sd, torch.cat((hflip(left), right), 0), mn2, mx2)
        mldisp, mrdisp = hflip(tdisp[0:B]), tdisp[B:]
    pan, disp, mask0, mask1 = falnet_forward(params, torch.cat((left, hflip(right)), 0), mn2, mx2,
                                             ret_disp=True, ret_pan=True, ret_subocc=True)  # :267-271
    rpan, lpan = pan[0:B], hflip(pan[B:])
    ldisp, rdisp = disp[0:B], hflip(disp[B:])
    lmask, rmask = mask0[0:B], hflip(mask0[B:])
    rlmask, lrmask = mask1[0:B], hflip(mask1[B:])
    with torch.no_grad():
        vgg_right, vgg_left = vgg_forward(vsd, right), vgg_forward(vsd, left)  # :289-291
    c2, c8 = int(0.20 * W), int(0.80 * W)
    O_L = (lmask * lrmask).clone()
    O_L[:, :, :, 0:c2] = 1  # :296-297
    O_R = (rmask * rlmask).clone()
    O_R[:, :, :, c8:] = 1  # :298-299
    rec = (rec_loss_fnc(vsd, O_R, rpan, right, vgg_right, a_p) +
           rec_loss_fnc(vsd, O_L, lpan, left, vgg_left, a_p)) / 2  # :304-305
    sm = (smoothness(left[:, :, :, c2:], ldisp[:, :, :, c2:], gamma=2) +
          smoothness(right[:, :, :, 0:c8], rdisp[:, :, :, 0:c8], gamma=2)) / 2  # :312-313
    nmaxl = 1 / F.max_pool2d(mldisp, kernel_size=(H, W))  # :319-320
    nmaxr = 1 / F.max_pool2d(mrdisp, kernel_size=(H, W))
    mirror = (torch.mean(nmaxl * (1 - O_L)[:, :, :, c2:] * torch.abs(ldisp - mldisp)[:, :, :, c2:]) +
              torch.mean(nmaxr * (1 - O_R)[:, :, :, 0:c8] * torch.abs(rdisp - mrdisp)[:, :, :, 0:c8])) / 2
    loss = rec + a_sm * sm + a_mr * mirror  # :327
    return {"loss": loss, "rec": rec, "sm": sm, "mirror": mirror, "ldisp": ldisp, "rdisp": rdisp,
            "rpan": rpan, "lpan": lpan, "O_L": O_L, "O_R": O_R}


# --------------------------------------------------------------------------- inference post-processing + metrics
def ms_pp(sd, left, disp, min_disp, max_disp):
    """Multi-scale/flip post-processing, Test_KITTI.py:287-300."""
    B, C, H, W = left.shape
    up = F.interpolate(hflip(left), scale_factor=2 / 3, mode="bilinear", align_corners=True)
    d2 = falnet_forward(sd, up, min_disp, max_disp)
    d2 = 1.5 * F.interpolate(d2, size=(H, W), mode="nearest")
    d2 = hflip(d2)
    norm = disp / (np.percentile(disp.detach().cpu().numpy(), 95) + 1e-6)
    norm = norm.clamp(max=1.0)
    return (1 - norm) * disp + norm * d2


def compute_kitti_errors(gt, pred, min_d=1.0, max_d=80.0):
    """myUtils.py:196-231 (use_median=False). numpy in, list of 7 floats out."""
    mask = gt > 0
    gt, pred = gt[mask].copy(), pred[mask].copy()
    pred = np.clip(pred, min_d, max_d)
    gt = np.clip(gt, min_d, max_d)
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = [(thresh < 1.25 ** i).mean() for i in (1, 2, 3)]
    rmse = math.sqrt(((gt - pred) ** 2).mean())
    rmse_log = math.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return [abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3]


def disp_to_depth(disp, focal=721.5377, baseline=0.54):
    """depth = f*b/disp with the zero-disparity guard of myUtils.py:234-253."""
    m = disp > 0
    return focal * baseline / (disp + (1.0 - m))
