"""CPU oracle for the CCVPE inference forward pass - TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (ccvpe_amd.models -> libccvpe_hip.so) never does and fails loudly without its HIP
library.

This is an independent fp32 PyTorch-CPU restatement of the reference algorithm, structured our
way (functional, state_dict-driven, aerial descriptor map as one k2/s2 convolution, rolling match
as a gather-correlation).  Each function cites the reference lines it follows.  Pinning:
tests/test_oracle_vs_reference.py compares it tensor-for-tensor with the reference imported from
/root/reference (in the build container only), and tests/test_oracle_golden.py checks it against
the committed golden vectors under tests/golden/ that oracle/make_golden.py captured from the
reference itself.  The reference ships no tests or golden vectors of its own.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from ccvpe_amd import spec


# ------------------------------------------------------------------------------------------
# EfficientNet-B0 encoder (efficientnet_pytorch/model.py:278-326, MBConvBlock.forward :90-131)
# ------------------------------------------------------------------------------------------
def _pad_static(x: torch.Tensor, k: int, s: int, circular: bool) -> torch.Tensor:
    """Conv2dStaticSamePadding / Conv2dStaticCircularPadding (utils.py:254-282, 330-358):
    amounts fixed by (k, s); circular mode wraps horizontally and zero-pads vertically."""
    lo, hi = spec.static_pad(k, s)
    if lo == 0 and hi == 0:
        return x
    if circular:
        x = F.pad(x, [lo, hi, 0, 0], mode="circular")
        return F.pad(x, [0, 0, lo, hi])
    return F.pad(x, [lo, hi, lo, hi])


def _bn(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    """Eval-mode BatchNorm2d, eps 1e-3 (utils.py:666)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=False, eps=spec.BN_EPS)


def _swish(x: torch.Tensor) -> torch.Tensor:
    """MemoryEfficientSwish forward (utils.py:64-80)."""
    return x * torch.sigmoid(x)


def mbconv(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, blk, circular: bool) -> torch.Tensor:
    """MBConvBlock.forward (model.py:90-131), eval mode (drop_connect is identity, utils.py:142)."""
    e, k, s, cin, cout = blk
    inp = x
    if e != 1:
        x = _swish(_bn(F.conv2d(x, sd[p + "._expand_conv.weight"]), sd, p + "._bn0"))
    x = _pad_static(x, k, s, circular)
    x = F.conv2d(x, sd[p + "._depthwise_conv.weight"], stride=s, groups=x.shape[1])
    x = _swish(_bn(x, sd, p + "._bn1"))
    sq = x.mean(dim=(2, 3), keepdim=True)
    sq = _swish(F.conv2d(sq, sd[p + "._se_reduce.weight"], sd[p + "._se_reduce.bias"]))
    sq = F.conv2d(sq, sd[p + "._se_expand.weight"], sd[p + "._se_expand.bias"])
    x = torch.sigmoid(sq) * x
    x = _bn(F.conv2d(x, sd[p + "._project_conv.weight"]), sd, p + "._bn2")
    if s == 1 and cin == cout:
        x = x + inp
    return x


def encoder(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, circular: bool
            ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """extract_features_multiscale (model.py:303-326): head volume + all 16 block outputs."""
    x = _pad_static(x, 3, 2, circular)
    x = _swish(_bn(F.conv2d(x, sd[p + "._conv_stem.weight"], stride=2), sd, p + "._bn0"))
    taps = []
    for i, blk in enumerate(spec.B0_BLOCKS):
        x = mbconv(x, sd, f"{p}._blocks.{i}", blk, circular)
        taps.append(x)
    x = _swish(_bn(F.conv2d(x, sd[p + "._conv_head.weight"]), sd, p + "._bn1"))
    return x, taps


# ------------------------------------------------------------------------------------------
# Descriptors
# ------------------------------------------------------------------------------------------
def ground_descriptor(vol: torch.Tensor, sd: Dict[str, torch.Tensor], level: int) -> torch.Tensor:
    """grd_feature_to_descriptorK (models.py:355-395): Conv2d(1280,c,1) -> permute(0,2,3,1) ->
    Conv2d(H_f,1,1) -> flatten.  Result layout d[w*c + ch]."""
    p = f"grd_feature_to_descriptor{level}"
    y = F.conv2d(vol, sd[p + ".0.weight"], sd[p + ".0.bias"])          # [B,c,Hf,Wf]
    wh = sd[p + ".2.weight"].reshape(-1)                                # [Hf]
    d = torch.einsum("bchw,h->bwc", y, wh) + sd[p + ".2.bias"].reshape(())
    return d.reshape(d.shape[0], -1)


def aerial_descriptor_map(vol: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """The 8x8 chunk loop of Linear(5120, D) (models.py:471-482) is one conv with kernel 2 stride 2:
    Flatten orders a [1280,2,2] patch as ch*4 + dy*2 + dx, i.e. W.view(D,1280,2,2)."""
    w = sd["sat_feature_to_descriptors.1.weight"]
    D = w.shape[0]
    return F.conv2d(vol, w.view(D, spec.HEAD_CH, 2, 2), sd["sat_feature_to_descriptors.1.bias"], stride=2)


# ------------------------------------------------------------------------------------------
# Rolling matching (models.py:485-511 and the five later copies)
# ------------------------------------------------------------------------------------------
def rolling_match(x: torch.Tensor, g: torch.Tensor, shifts: Sequence[int]) -> torch.Tensor:
    """score_r(p) = sum_c g[c] * x[(c+s_r) mod C, p] / (||x[(c+s_r) mod C, p]||_c * ||g||), no epsilon
    (models.py:494).  x [B,C,H,W], g [B,L] -> [B,R,H,W]."""
    B, C, H, W = x.shape
    L = g.shape[1]
    c = torch.arange(L)
    gn = g.norm(dim=1).view(B, 1, 1)
    out = []
    for s in shifts:
        win = x[:, (c + s) % C]                                   # [B,L,H,W]
        num = torch.einsum("blhw,bl->bhw", win, g)
        den = win.norm(dim=1) * gn
        out.append(num / den)
    return torch.stack(out, dim=1)


def _decoder_level(x: torch.Tensor, skip: Optional[torch.Tensor], sd: Dict[str, torch.Tensor], n: int, sfx: str
                   ) -> torch.Tensor:
    """deconvN -> cat(skip) -> convN (double_conv, models.py:42-47, 516-518)."""
    x = F.conv_transpose2d(x, sd[f"deconv{n}{sfx}.weight"], sd[f"deconv{n}{sfx}.bias"], stride=2)
    if skip is not None:
        x = torch.cat([x, skip], dim=1)
    x = F.relu(F.conv2d(x, sd[f"conv{n}{sfx}.0.weight"], sd[f"conv{n}{sfx}.0.bias"], padding=1))
    return F.conv2d(x, sd[f"conv{n}{sfx}.2.weight"], sd[f"conv{n}{sfx}.2.bias"], padding=1)


def forward(variant: str, sd: Dict[str, torch.Tensor], grd: torch.Tensor, sat: torch.Tensor,
            circular: bool = False, ori_noise: Optional[float] = None, taps: Optional[dict] = None, grad: bool = False):
    """CVM_*.forward (models.py:150-343, 448-652, 752-950, 1051-1244).

    Returns the reference 9-tuple (logits_flattened, heatmap, x_ori, ms1..ms6).  `taps`, if a dict,
    receives intermediate tensors for mismatch localisation.  `grad=True` keeps autograd recording (the reference
    test loop calls the model without no_grad, train_VIGOR.py:282) - only the timing in bench.py's cpu_baseline uses it.
    """
    v = spec.VARIANTS[variant]
    if variant == "vigor_ori_prior":
        assert ori_noise is not None
    with torch.set_grad_enabled(grad):
        gvol, _ = encoder(grd, sd, "grd_efficientnet", circular)
        descs = [ground_descriptor(gvol, sd, k) for k in range(1, 7)]
        svol, blocks = encoder(sat, sd, "sat_efficientnet", False)
        skips = [blocks[i] for i in spec.TAP_BLOCKS]
        dmap = aerial_descriptor_map(svol, sd)
        if taps is not None:
            taps["grd_volume"] = gvol
            taps["sat_volume"] = svol
            for k, d in enumerate(descs, 1):
                taps[f"grd_desc{k}"] = d
            for i, t in zip(spec.TAP_BLOCKS, skips):
                taps[f"sat_block{i}"] = t
            taps["sat_descriptor_map"] = dmap

        ms = []
        x = dmap
        ms1_full = None
        for k in range(1, 7):
            g = descs[k - 1]
            L = g.shape[1]
            sh = spec.roll_shifts(v, k, L, ori_noise)
            score = rolling_match(x, g, sh)
            if k == 1:
                # level-1 stack fed to the orientation decoder is always the full roll set
                # (models.py:501-511); it is also what the 9-tuple returns as ms1.
                ms1_full = rolling_match(x, g, spec.full_roll_shifts(v, 1, L)) if variant == "vigor_ori_prior" else score
                ms.append(ms1_full)
            else:
                ms.append(score)
            smax = score.max(dim=1, keepdim=True).values
            x = torch.cat([smax, F.normalize(x, p=2, dim=1)], dim=1)
            n = 7 - k
            skip = skips[k - 1] if k <= 5 else None
            x = _decoder_level(x, skip, sd, n, "")
            if taps is not None:
                taps[f"loc_level{n}"] = x
        logits = x.flatten(1)
        heat = torch.softmax(logits, dim=-1).reshape(x.shape)

        xo = torch.cat([ms1_full, F.normalize(dmap, p=2, dim=1)], dim=1)
        for k in range(1, 7):
            n = 7 - k
            skip = skips[k - 1] if k <= 5 else None
            xo = _decoder_level(xo, skip, sd, n, "_ori")
            if taps is not None:
                taps[f"ori_level{n}"] = xo
        xo = F.normalize(xo, p=2, dim=1)
    return (logits, heat, xo, *ms)


def postprocess(heatmap: torch.Tensor, ori: torch.Tensor):
    """Test-loop post-processing (train_VIGOR.py:297-316): argmax of the heatmap, the (cos,sin) at that
    pixel, and the angle in degrees.  Returns (flat_index [B], prob [B], cos [B], sin [B], deg [B])."""
    B = heatmap.shape[0]
    flat = heatmap.reshape(B, -1)
    prob, idx = flat.max(dim=1)
    W = heatmap.shape[-1]
    yy, xx = idx // W, idx % W
    cs = ori[torch.arange(B), 0, yy, xx]
    sn = ori[torch.arange(B), 1, yy, xx]
    a = torch.rad2deg(torch.acos(cs.clamp(-1, 1)))
    deg = torch.where(sn < 0, (-a) % 360, a)
    return idx, prob, cs, sn, deg


def eval_metrics(heatmap, ori, gt_index, meter_per_pixel, gt_cos_sin=None, heading_deg=None):
    """Ground-truth side of the test loops, restated line by line with the same numpy / math calls
    (train_VIGOR.py:296-326; lateral / longitudinal split train_KITTI.py:318-325).  The drivers are scripts, not functions,
    and need the datasets, so this restatement is parity-unpinned beyond its line-by-line correspondence.
    heatmap [B,1,H,W], ori [B,2,H,W] numpy float32; gt_index [B] flat argmax of the GT map.  Returns dict of float64 lists
    (NaN where the reference appends nothing)."""
    import math
    import numpy as np
    B = heatmap.shape[0]
    W = heatmap.shape[-1]
    keys = ("pixel_distance", "meter_distance", "prob_at_gt", "angle_pred_deg", "angle_gt_deg", "orientation_error_deg", "longitudinal_m", "lateral_m")
    out = {k: np.full(B, np.nan) for k in keys}
    mpp = np.broadcast_to(np.asarray(meter_per_pixel, dtype=np.float64), (B,))
    for b in range(B):
        current_pred = heatmap[b]
        loc_pred = np.unravel_index(current_pred.argmax(), current_pred.shape)
        loc_gt = (0, int(gt_index[b]) // W, int(gt_index[b]) % W)
        pixel_distance = np.sqrt((loc_gt[1] - loc_pred[1]) ** 2 + (loc_gt[2] - loc_pred[2]) ** 2)
        out["pixel_distance"][b] = pixel_distance
        out["meter_distance"][b] = pixel_distance * mpp[b]
        cos_pred, sin_pred = ori[b, :, loc_pred[1], loc_pred[2]]
        if np.abs(cos_pred) <= 1 and np.abs(sin_pred) <= 1:
            a_acos_pred = math.acos(cos_pred)
            angle_pred = math.degrees(-a_acos_pred) % 360 if sin_pred < 0 else math.degrees(a_acos_pred)
            out["angle_pred_deg"][b] = angle_pred
            if gt_cos_sin is not None:
                cos_gt, sin_gt = gt_cos_sin[b]
                a_acos_gt = math.acos(cos_gt)
                angle_gt = math.degrees(-a_acos_gt) % 360 if sin_gt < 0 else math.degrees(a_acos_gt)
                out["angle_gt_deg"][b] = angle_gt
                out["orientation_error_deg"][b] = np.min([np.abs(angle_gt - angle_pred), 360 - np.abs(angle_gt - angle_pred)])
        out["prob_at_gt"][b] = heatmap[b, 0, loc_gt[1], loc_gt[2]]
        if heading_deg is not None:
            gt2pred_from_north = np.arctan2(np.abs(loc_gt[2] - loc_pred[2]), np.abs(loc_gt[1] - loc_pred[1])) * 180 / math.pi
            angle_diff = np.abs(heading_deg[b] - gt2pred_from_north)
            out["longitudinal_m"][b] = np.abs(np.cos(angle_diff * np.pi / 180) * pixel_distance) * mpp[b]
            out["lateral_m"][b] = np.abs(np.sin(angle_diff * np.pi / 180) * pixel_distance) * mpp[b]
    return out


def preprocess(img_u8_hwc, shift=None, crop_w=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """CPU restatement of the input pipeline after decode/resize: ToTensor + Normalize
    (train_VIGOR.py:57-70), panorama roll torch.roll(grd, shift, dims=2) (datasets.py:118), FoV width crop
    grd[..., :crop_w] (train_VIGOR.py:272-273).  img_u8_hwc: uint8 tensor [B,H,W,3]."""
    x = img_u8_hwc.permute(0, 3, 1, 2).to(torch.float32) / 255.0                 # ToTensor
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    x = (x - m) / s                                                              # Normalize
    if shift is not None:
        x = torch.stack([torch.roll(x[b], int(shift[b]), dims=2) for b in range(x.shape[0])])
    if crop_w is not None:
        x = x[..., :crop_w]
    return x.contiguous()
