"""Kernel-level parity of the implicit-GEMM convolution (through the C ABI hook ccvpe_op_conv2d) against
torch's fp32 conv on ragged / odd shapes: M and N not multiples of any tile, every tile id, strides,
the 2x2 s2 form of the aerial descriptor map, activations.  Tolerance 2e-5 of the output scale."""
import pytest
import torch
import torch.nn.functional as F

from ccvpe_amd import _lib

pytestmark = pytest.mark.gpu

SHAPES = [
    # B, H, W, Cin, Cout, K, stride, pad
    (1, 5, 7, 8, 3, 3, 1, 1),
    (2, 9, 9, 16, 17, 1, 1, 0),
    (1, 16, 16, 24, 40, 3, 1, 1),
    (2, 8, 8, 64, 126, 1, 1, 0),
    (1, 12, 10, 8, 8, 2, 2, 0),
    (3, 16, 16, 1280, 64, 2, 2, 0),
    (1, 33, 17, 40, 96, 3, 1, 1),
    (2, 31, 29, 104, 80, 3, 1, 1),
    (1, 64, 64, 16, 16, 3, 1, 1),
    (1, 7, 4, 320, 1280, 1, 1, 0),
]


def ref_conv(x, w, b, stride, pad, act):
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=stride, padding=pad)
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = y * torch.sigmoid(y)
    return y.permute(0, 2, 3, 1).float()


@pytest.mark.parametrize("shape", SHAPES)
def test_conv2d_matches_torch_auto_tile(shape):
    B, H, W, Cin, Cout, K, stride, pad = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, K, K, device="cuda", generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    for act in (0, 1, 2):
        out, _ = _lib.op_conv2d(x, w, b, stride, pad, act, 0)
        ref = ref_conv(x, w, b, stride, pad, act)
        assert out.shape == ref.shape
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-5, f"act {act}: {err:.3g}"


def test_every_tile_config_is_correct():
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(2, 19, 23, 40, device="cuda", generator=g)     # M = 874: ragged for every BM
    w = torch.randn(88, 40, 3, 3, device="cuda", generator=g) / 19.0
    b = torch.randn(88, device="cuda", generator=g)
    ref = ref_conv(x, w, b, 1, 1, 0)
    n = lib.ccvpe_op_num_tiles()
    assert n >= 8
    for t in range(1, n + 1):
        name = lib.ccvpe_op_tile_name(t).decode()
        if "wino" in name:       # 19 x 23 is not Winograd-shaped: covered by test_winograd_*
            with pytest.raises(_lib.CcvpeError):
                _lib.op_conv2d(x, w, b, 1, 1, 0, t)
            continue
        out, _ = _lib.op_conv2d(x, w, b, 1, 1, 0, t)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        tol = 1e-4 if "bf16x3" in name else 2e-5      # the 3-term bf16 split carries ~2^-16 per product
        assert err <= tol, f"tile {t} ({name}): {err:.3g}"


X_WIDTHS = [32, 48, 64, 80, 96, 128]     # conv_wino4x_<width>: the xi-split F(4x4) form takes a layer in ONE n-block of its width


def x_tile_applies(name, cout):
    """conv_wino4x_<w> serves exactly the layers whose narrowest fitting configuration it is (24 <= Cout <= w)."""
    w = int(name.rsplit("_", 1)[1])
    return cout >= 24 and next((x for x in X_WIDTHS if cout <= x), None) == w


WINO_SHAPES = [
    # B, H, W, Cin, Cout
    (1, 16, 16, 8, 16),
    (3, 16, 16, 64, 88),
    (2, 48, 32, 24, 40),
    (1, 32, 48, 104, 17),      # not a multiple of 4 channels: every Winograd tile refuses it
    (1, 32, 48, 104, 20),      # a partly filled last 16-channel slice
    (2, 16, 16, 200, 160),
    (1, 64, 64, 16, 100),
    (1, 32, 32, 48, 32),       # conv2_ori-shaped: the 32-wide xi-split configuration
    (2, 16, 32, 88, 64),       # conv3_ori-shaped (64), Cin = 88: a half-filled last 16-channel group
    (1, 32, 32, 104, 80),      # conv3-shaped: eight waves, 3 + 2 slices
    (1, 16, 16, 24, 30),       # not a multiple of 4 channels: only the xi-split form (one channel per lane in its epilogue) takes it
]


@pytest.mark.parametrize("shape", WINO_SHAPES)
def test_winograd_tiles_match_torch(shape):
    """Winograd F(2x2,3x3) and F(4x4,3x3) kernels (decoder double_conv, models.py:42-47): every variant, ragged Cout,
    bias + ReLU epilogue, halo handling at all four image borders, against an fp64 torch convolution."""
    lib = _lib.load()
    B, H, W, Cin, Cout = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    tiles = [t for t in range(1, lib.ccvpe_op_num_tiles() + 1) if b"wino" in lib.ccvpe_op_tile_name(t)]
    assert len(tiles) >= 3
    for act in (0, 1):
        ref = ref_conv(x, w, b, 1, 1, act)
        for t in tiles:
            name = lib.ccvpe_op_tile_name(t).decode()
            f4 = "wino4" in name
            if "wino4x" in name:
                refuse = not x_tile_applies(name, Cout)
            else:
                refuse = (f4 and Cout < 40) or Cout % 4 != 0   # F(4x4,3x3) weights are only packed for layers of >= 40 output channels; both forms store 4 channels per lane
            if refuse:
                with pytest.raises(_lib.CcvpeError):
                    _lib.op_conv2d(x, w, b, 1, 1, act, t)
                continue
            out, _ = _lib.op_conv2d(x, w, b, 1, 1, act, t)
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            # F(2x2): transforms only add / subtract (measured 2-12e-7).  F(4x4): constants up to 8 amplify the fp32 rounding
            # of the transforms (~1.4e-5 expected); both far inside the 1e-3 contract
            assert err <= (1e-4 if f4 else 2e-5), f"tile {name} act {act}: {err:.3g}"


@pytest.mark.parametrize("shape,splitk", [((2, 16, 16, 200, 160), 2), ((1, 32, 48, 104, 100), 4), ((3, 16, 16, 64, 88), 2), ((1, 16, 16, 104, 40), 3),
                                          # more slices than equal shares fill: 5 / 3 groups of 16 channels, 10 / 5 chunks of 8 over 4 slices
                                          ((1, 16, 16, 80, 64), 4), ((1, 16, 16, 40, 64), 4), ((2, 16, 16, 1344, 64), 16)])
def test_winograd_split_k(shape, splitk):
    """Split-K form of the Winograd tiles (slab + reduce), incl. a K that does not divide into equal slices and the
    16-channel group tail of the F(4x4,3x3) kernel (Cin = 200, 104)."""
    lib = _lib.load()
    B, H, W, Cin, Cout = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape) + splitk)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    ref = ref_conv(x, w, b, 1, 1, 1)
    for t in range(1, lib.ccvpe_op_num_tiles() + 1):
        name = lib.ccvpe_op_tile_name(t).decode()
        if "wino" not in name or ("wino4x" in name and not x_tile_applies(name, Cout)):
            continue
        out, _ = _lib.op_conv2d(x, w, b, 1, 1, 1, t | (splitk << 8))
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        assert err <= (1e-4 if "wino4" in name else 2e-5), f"tile {name} split-K {splitk}: {err:.3g}"
        # round 4: the same split reducing itself (split code 64 + S: write-through slabs, a ticket per output region, the last K slice
        # sums the slabs in slice order) - the same bits as slab + reduce launch; three runs: whoever arrives last, the same bits
        for _ in range(3):
            fused, _ = _lib.op_conv2d(x, w, b, 1, 1, 1, t | ((64 + splitk) << 8))
            assert torch.equal(fused, out), f"tile {name} self-reducing split-K {splitk}"


@pytest.mark.parametrize("shape,splitk", [((1, 16, 16, 1152, 192, 1, 1, 0), 8), ((1, 8, 8, 1288, 4096, 1, 1, 0), 4), ((1, 16, 16, 1280, 320, 2, 2, 0), 16),
                                          ((2, 10, 20, 1280, 126, 1, 1, 0), 8), ((1, 16, 16, 1344, 640, 3, 1, 1), 12)])
def test_implicit_gemm_self_reducing_split_k(shape, splitk):
    """conv_igemm_kernel with split code 64 + S against slab + reduce launch (same bits) and torch: deep-K 1x1, a transposed-conv-shaped
    GEMM, the k2s2 descriptor conv, a width that is not a multiple of 4 (element-wise slab path), a 3x3 decoder conv at batch 1."""
    lib = _lib.load()
    B, H, W, Cin, Cout, K, stride, pad = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape) + splitk)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, K, K, device="cuda", generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    ref = ref_conv(x, w, b, stride, pad, 2)
    for name in ("conv_igemm_64x32_m16", "conv_igemm_64x64_m32_s1", "conv_igemm_128x128_m16"):
        t = _tile_id(name)
        out, _ = _lib.op_conv2d(x, w, b, stride, pad, 2, t | (splitk << 8))
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-5, f"{name}: {err:.3g}"
        for _ in range(3):
            fused, _ = _lib.op_conv2d(x, w, b, stride, pad, 2, t | ((64 + splitk) << 8))
            assert torch.equal(fused, out), name


def test_winograd_tail_split():
    """cfg split code 255 of the F(4x4,3x3) tiles: 2 x 16 x 128 pixels = 16 pixel blocks x 33 channel blocks of 64 = 528 work items on
    512 resident workgroups -> 512 run whole, the last channel block (16 items) is split over K through slabs (the same grid
    arithmetic as conv5.0 at batch 32: 640 items); the 128-channel form has no whole-block remainder and runs unsplit."""
    lib = _lib.load()
    B, H, W, Cin, Cout = 2, 16, 128, 64, 2112
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    for act in (0, 1):
        ref = ref_conv(x, w, b, 1, 1, act)
        for t in range(1, lib.ccvpe_op_num_tiles() + 1):
            name = lib.ccvpe_op_tile_name(t).decode()
            if "wino4" not in name or "wino4x" in name:
                continue
            out, _ = _lib.op_conv2d(x, w, b, 1, 1, act, t | (255 << 8))
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 1e-4, f"tile {name} tail split act {act}: {err:.3g}"


PW_SHAPES = [
    # B, H, W, Cin, Cout: 1x1 layers of the encoders (expand / project / head shapes, ragged M and N, K tails)
    (2, 16, 16, 16, 96), (1, 19, 23, 40, 240), (3, 8, 8, 112, 672), (2, 16, 16, 480, 80), (1, 16, 16, 24, 144), (2, 5, 7, 328, 40),
]


@pytest.mark.parametrize("shape", PW_SHAPES)
def test_pointwise_persistent_tiles_match_torch(shape):
    """conv_pw_kernel (1x1 convs with K <= 512 straight from HBM into the MFMA operand registers): every column width,
    swish / none epilogues, rows and columns that do not fill the last tile, K that is not a multiple of 16."""
    lib = _lib.load()
    B, H, W, Cin, Cout = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 1, 1, device="cuda", generator=g) / Cin ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    tiles = [t for t in range(1, lib.ccvpe_op_num_tiles() + 1) if b"conv_pw" in lib.ccvpe_op_tile_name(t)]
    assert len(tiles) >= 4
    for act in (0, 2):
        ref = ref_conv(x, w, b, 1, 0, act)
        for t in tiles:
            out, _ = _lib.op_conv2d(x, w, b, 1, 0, act, t)
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 2e-5, f"tile {lib.ccvpe_op_tile_name(t).decode()} act {act}: {err:.3g}"


def test_conv2d_rejects_bad_geometry():
    x = torch.randn(1, 4, 4, 12, device="cuda")
    w = torch.randn(4, 12, 1, 1, device="cuda")
    with pytest.raises(_lib.CcvpeError):
        _lib.op_conv2d(x, w)              # Cin not a multiple of 8
    x = torch.randn(1, 8, 8, 8, device="cuda")
    w = torch.randn(4, 8, 5, 5, device="cuda")
    with pytest.raises(_lib.CcvpeError):
        _lib.op_conv2d(x, w, pad=2)       # 25 taps: outside the kernel's tap table


def _tile_id(name):
    lib = _lib.load()
    for t in range(1, lib.ccvpe_op_num_tiles() + 1):
        if lib.ccvpe_op_tile_name(t).decode() == name:
            return t
    raise KeyError(name)


PROJL_SHAPES = [
    # B, H, W, Cin, Cout, K, stride   (1x1 / k2s2, pad 0): the latency form of kernels_proj.hip
    (1, 16, 16, 1152, 192, 1, 1),     # block 12 project (gate-less through this hook): 72 steps -> 4-5 per wave
    (1, 5, 8, 1152, 320, 1, 1),       # Oxford's 40-row map: ragged row tile
    (2, 10, 20, 1280, 126, 1, 1),     # ground descriptor heads: N not a multiple of 16
    (1, 8, 8, 1288, 4096, 1, 1),      # loc6 transposed conv as its 1x1 GEMM: K not a multiple of 16 (81 steps: two groups per wave)
    (1, 16, 16, 1280, 1280, 2, 2),    # aerial descriptor map: four taps, 320 steps (ten per group)
    (3, 8, 8, 480, 80, 1, 1),         # 30 steps: waves without work
]


@pytest.mark.parametrize("shape", PROJL_SHAPES)
def test_latency_form_deep_k_gemm_matches_torch(shape):
    """conv_proj_lat_kernel through ccvpe_op_conv2d: sixteen waves split K, partial sums meet in LDS in wave order."""
    B, H, W, Cin, Cout, K, stride = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, K, K, device="cuda", generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    ref = ref_conv(x, w, b, stride, 0, 0)
    # (..._r2 / _r4: two / four row tiles per workgroup - the weights of a 64-row layer are read once)
    for name in ("conv_projl_1", "conv_projl_2", "conv_projl_4", "conv_projl_r2", "conv_projl_r4"):
        out, _ = _lib.op_conv2d(x, w, b, stride, 0, 0, _tile_id(name))   # (fewer rows than a workgroup takes: the library's own pick runs)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-5, f"{name}: {err:.3g}"
        again, _ = _lib.op_conv2d(x, w, b, stride, 0, 0, _tile_id(name))
        assert torch.equal(out, again)
    # self-reducing split-K of the latency form (split code 64 + S): K slices on gridDim.z, slabs + ticket, the last slice sums in slice order
    for name, S in (("conv_projl_1", 2), ("conv_projl_r4", 4), ("conv_projl_2", 8)):
        out, _ = _lib.op_conv2d(x, w, b, stride, 0, 0, _tile_id(name) | ((64 + S) << 8))
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-5, f"{name} split {S}: {err:.3g}"
        for _ in range(3):
            again, _ = _lib.op_conv2d(x, w, b, stride, 0, 0, _tile_id(name) | ((64 + S) << 8))
            assert torch.equal(out, again)
