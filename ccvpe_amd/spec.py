"""Static description of the CCVPE inference path: variants, layer widths, state_dict layout.

Everything here is plain data derived from reading the reference (file:line cited per item);
it is shared by the host-side mirror (`ccvpe_amd.models`), the weight generator, the oracle
and the tests.  No torch import at module level - the C-ABI side only needs the numbers.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

# --------------------------------------------------------------------------------------------
# EfficientNet-B0 as CCVPE instantiates it.
# Block schedule: reference efficientnet_pytorch/utils.py:647-655 (block strings), expanded per
# model.py:187-203 (first repeat carries the stride / channel change, later repeats stride 1).
# (expand_ratio, kernel, stride, cin, cout)
# --------------------------------------------------------------------------------------------
B0_BLOCKS: Tuple[Tuple[int, int, int, int, int], ...] = (
    (1, 3, 1, 32, 16),
    (6, 3, 2, 16, 24), (6, 3, 1, 24, 24),
    (6, 5, 2, 24, 40), (6, 5, 1, 40, 40),
    (6, 3, 2, 40, 80), (6, 3, 1, 80, 80), (6, 3, 1, 80, 80),
    (6, 5, 1, 80, 112), (6, 5, 1, 112, 112), (6, 5, 1, 112, 112),
    (6, 5, 2, 112, 192), (6, 5, 1, 192, 192), (6, 5, 1, 192, 192), (6, 5, 1, 192, 192),
    (6, 3, 1, 192, 320),
)
STEM_CH = 32          # model.py:180
HEAD_CH = 1280        # model.py:207
BN_EPS = 1e-3         # utils.py:666
SE_RATIO = 0.25       # utils.py:647-655 (se0.25)
FC_CLASSES = 1000     # model.py:216, unused by the CCVPE forward but part of the state_dict
# encoder block outputs the decoders concatenate (models.py:465-469)
TAP_BLOCKS = (15, 10, 4, 2, 0)   # level 6 .. level 2 skip connections
TAP_CH = (320, 112, 40, 24, 16)


def se_squeeze(cin: int) -> int:
    """model.py:79 - max(1, int(input_filters * se_ratio))."""
    return max(1, int(cin * SE_RATIO))


def static_pad(k: int, s: int) -> Tuple[int, int]:
    """(before, after) padding of Conv2dStatic{Same,Circular}Padding (utils.py:261-277, 337-351).

    The reference derives the amounts at construction from the nominal 224-pixel image walked
    through the strides (model.py:175-203); every nominal size is even where a stride-2 conv
    sits, so the amounts depend on (k, s) only:  s=1 -> (k-1)/2 both sides;  s=2 -> total k-2,
    split floor/ceil.
    """
    if s == 1:
        return ((k - 1) // 2, (k - 1) // 2)
    total = k - 2
    return (total // 2, total - total // 2)


def conv_out(n: int, k: int, s: int) -> int:
    lo, hi = static_pad(k, s)
    return (n + lo + hi - k) // s + 1


def encoder_shapes(h: int, w: int) -> List[Tuple[int, int]]:
    """Spatial size after the stem and after each of the 16 blocks (index 0 = stem)."""
    out = [(conv_out(h, 3, 2), conv_out(w, 3, 2))]
    for (_, k, s, _, _) in B0_BLOCKS:
        ph, pw = out[-1]
        out.append((conv_out(ph, k, s), conv_out(pw, k, s)))
    return out


@dataclass(frozen=True)
class DecoderLevel:
    """One decoder level: ConvTranspose2d(k2,s2) -> cat(skip) -> conv3x3+ReLU -> conv3x3."""
    deconv_in: int
    deconv_out: int
    skip: int          # 0 at the last level (no encoder tap)
    mid: int           # double_conv width (both convs output `mid` except the last level)
    out: int


@dataclass(frozen=True)
class Variant:
    name: str
    cls_name: str
    grd_hw: Tuple[int, int]                 # nominal ground image size (full FoV)
    feat_h: int                             # rows of the ground feature volume = Conv2d(feat_h,1,1) in the heads
    head_ch: Tuple[int, ...]                # 1x1 conv widths of the 6 ground heads (level 1..6)
    sat_desc: int                           # Linear(5120, D)
    match_ch: Tuple[int, ...]               # channels C_k of the aerial tensor matched at level k
    step: Tuple[int, ...]                   # roll step per level
    n_rolls: int                            # full roll count (20 VIGOR/Oxford, 16 KITTI)
    centre_window: bool                     # Oxford: window taken from the centre of the rolled tensor
    loc: Tuple[DecoderLevel, ...]           # level 6 .. level 1
    ori: Tuple[DecoderLevel, ...]


def _vigor_dec() -> Tuple[Tuple[DecoderLevel, ...], Tuple[DecoderLevel, ...]]:
    # models.py:407-446 (identical in CVM_VIGOR :109-148 and CVM_OxfordRobotCar :1009-1048)
    loc = (
        DecoderLevel(1281, 1024, 320, 640, 640),
        DecoderLevel(641, 320, 112, 320, 320),
        DecoderLevel(321, 160, 40, 160, 160),
        DecoderLevel(161, 80, 24, 80, 80),
        DecoderLevel(81, 40, 16, 40, 40),
        DecoderLevel(41, 16, 0, 16, 1),
    )
    ori = (
        DecoderLevel(1300, 1024, 320, 640, 640),
        DecoderLevel(640, 256, 112, 256, 256),
        DecoderLevel(256, 128, 40, 128, 128),
        DecoderLevel(128, 64, 24, 64, 64),
        DecoderLevel(64, 32, 16, 32, 32),
        DecoderLevel(32, 16, 0, 16, 2),
    )
    return loc, ori


def _kitti_dec() -> Tuple[Tuple[DecoderLevel, ...], Tuple[DecoderLevel, ...]]:
    # models.py:710-749
    loc = (
        DecoderLevel(2049, 1024, 320, 512, 512),
        DecoderLevel(513, 256, 112, 256, 256),
        DecoderLevel(257, 128, 40, 128, 128),
        DecoderLevel(129, 64, 24, 128, 128),      # conv3 widens to 128 (models.py:720)
        DecoderLevel(129, 32, 16, 32, 32),
        DecoderLevel(33, 16, 0, 16, 1),
    )
    ori = (
        DecoderLevel(2064, 1024, 320, 512, 512),
        DecoderLevel(512, 256, 112, 256, 256),
        DecoderLevel(256, 128, 40, 128, 128),
        DecoderLevel(128, 64, 24, 64, 64),
        DecoderLevel(64, 32, 16, 32, 32),
        DecoderLevel(32, 16, 0, 16, 2),
    )
    return loc, ori


_VL, _VO = _vigor_dec()
_KL, _KO = _kitti_dec()

VARIANTS: Dict[str, Variant] = {
    # models.py:49-343
    "vigor": Variant("vigor", "CVM_VIGOR", (320, 640), 10, (64, 32, 16, 8, 4, 2), 1280,
                     (1280, 640, 320, 160, 80, 40), (64, 32, 16, 8, 4, 2), 20, False, _VL, _VO),
    # models.py:346-652
    "vigor_ori_prior": Variant("vigor_ori_prior", "CVM_VIGOR_ori_prior", (320, 640), 10,
                               (64, 32, 16, 8, 4, 2), 1280, (1280, 640, 320, 160, 80, 40),
                               (64, 32, 16, 8, 4, 2), 20, False, _VL, _VO),
    # models.py:655-950
    "kitti": Variant("kitti", "CVM_KITTI", (256, 1024), 8, (16, 8, 4, 2, 1, 1), 2048,
                     (2048, 512, 256, 128, 128, 32), (128, 64, 32, 16, 8, 8), 16, False, _KL, _KO),
    # models.py:954-1244
    "oxford": Variant("oxford", "CVM_OxfordRobotCar", (154, 231), 4, (32, 16, 8, 4, 2, 1), 1280,
                      (1280, 640, 320, 160, 80, 40), (64, 32, 16, 8, 4, 2), 20, True, _VL, _VO),
}

SAT_HW = (512, 512)
OUT_HW = (512, 512)


def roll_shifts(v: Variant, level: int, desc_len: int, ori_noise: Optional[float]) -> List[int]:
    """Channel shifts s_r such that window_r[c] = x[(c + s_r) mod C], c in [0, desc_len).

    torch.roll(x, -i*step, dims=1)[:, c] == x[:, (c + i*step) mod C]  (models.py:192-193);
    the ori-prior variant iterates i = -n..n with n = int(ori_noise/18) (models.py:489-491);
    Oxford slices the rolled tensor at int(C/2 - L/2) (models.py:1094).
    `level` is 1-based.
    """
    C = v.match_ch[level - 1]
    step = v.step[level - 1]
    off = int(C / 2 - desc_len / 2) if v.centre_window else 0
    if v.name == "vigor_ori_prior":
        n = int(ori_noise / 18)
        idx = range(-n, n + 1)
    else:
        idx = range(v.n_rolls)
    return [(off + i * step) % C for i in idx]


def full_roll_shifts(v: Variant, level: int, desc_len: int) -> List[int]:
    """The always-full roll set (20 or 16) used for the level-1 stack fed to the orientation decoder
    (models.py:501-511, 632)."""
    C = v.match_ch[level - 1]
    step = v.step[level - 1]
    off = int(C / 2 - desc_len / 2) if v.centre_window else 0
    return [(off + i * step) % C for i in range(v.n_rolls)]


# --------------------------------------------------------------------------------------------
# state_dict layout (818 keys, identical key set for all four classes - SURVEY 8b)
# --------------------------------------------------------------------------------------------
def _bn_keys(prefix: str, c: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    return [
        (prefix + ".weight", (c,), "f32"),
        (prefix + ".bias", (c,), "f32"),
        (prefix + ".running_mean", (c,), "f32"),
        (prefix + ".running_var", (c,), "f32"),
        (prefix + ".num_batches_tracked", (), "i64"),
    ]


def encoder_keys(prefix: str) -> List[Tuple[str, Tuple[int, ...], str]]:
    ks: List[Tuple[str, Tuple[int, ...], str]] = []
    ks.append((f"{prefix}._conv_stem.weight", (STEM_CH, 3, 3, 3), "f32"))
    ks += _bn_keys(f"{prefix}._bn0", STEM_CH)
    for i, (e, k, s, cin, cout) in enumerate(B0_BLOCKS):
        p = f"{prefix}._blocks.{i}"
        mid = cin * e
        if e != 1:
            ks.append((f"{p}._expand_conv.weight", (mid, cin, 1, 1), "f32"))
            ks += _bn_keys(f"{p}._bn0", mid)
        ks.append((f"{p}._depthwise_conv.weight", (mid, 1, k, k), "f32"))
        ks += _bn_keys(f"{p}._bn1", mid)
        sq = se_squeeze(cin)
        ks.append((f"{p}._se_reduce.weight", (sq, mid, 1, 1), "f32"))
        ks.append((f"{p}._se_reduce.bias", (sq,), "f32"))
        ks.append((f"{p}._se_expand.weight", (mid, sq, 1, 1), "f32"))
        ks.append((f"{p}._se_expand.bias", (mid,), "f32"))
        ks.append((f"{p}._project_conv.weight", (cout, mid, 1, 1), "f32"))
        ks += _bn_keys(f"{p}._bn2", cout)
    ks.append((f"{prefix}._conv_head.weight", (HEAD_CH, B0_BLOCKS[-1][4], 1, 1), "f32"))
    ks += _bn_keys(f"{prefix}._bn1", HEAD_CH)
    ks.append((f"{prefix}._fc.weight", (FC_CLASSES, HEAD_CH), "f32"))
    ks.append((f"{prefix}._fc.bias", (FC_CLASSES,), "f32"))
    return ks


def state_dict_spec(v: Variant) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (key, shape, dtype) list in the reference's registration order."""
    ks = encoder_keys("grd_efficientnet")
    for lvl, c in enumerate(v.head_ch, 1):
        ks.append((f"grd_feature_to_descriptor{lvl}.0.weight", (c, HEAD_CH, 1, 1), "f32"))
        ks.append((f"grd_feature_to_descriptor{lvl}.0.bias", (c,), "f32"))
        ks.append((f"grd_feature_to_descriptor{lvl}.2.weight", (1, v.feat_h, 1, 1), "f32"))
        ks.append((f"grd_feature_to_descriptor{lvl}.2.bias", (1,), "f32"))
    ks += encoder_keys("sat_efficientnet")
    ks.append(("sat_feature_to_descriptors.1.weight", (v.sat_desc, HEAD_CH * 4), "f32"))
    ks.append(("sat_feature_to_descriptors.1.bias", (v.sat_desc,), "f32"))
    for suffix, dec in (("", v.loc), ("_ori", v.ori)):
        for j, lv in enumerate(dec):
            n = 6 - j
            ks.append((f"deconv{n}{suffix}.weight", (lv.deconv_in, lv.deconv_out, 2, 2), "f32"))
            ks.append((f"deconv{n}{suffix}.bias", (lv.deconv_out,), "f32"))
            cin = lv.deconv_out + lv.skip
            ks.append((f"conv{n}{suffix}.0.weight", (lv.mid, cin, 3, 3), "f32"))
            ks.append((f"conv{n}{suffix}.0.bias", (lv.mid,), "f32"))
            ks.append((f"conv{n}{suffix}.2.weight", (lv.out, lv.mid, 3, 3), "f32"))
            ks.append((f"conv{n}{suffix}.2.bias", (lv.out,), "f32"))
    return ks
