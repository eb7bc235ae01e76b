"""SURVEY 8f "next" rows built this round: device pre-processing (row 2) and aerial-side caching (row 4)."""
import numpy as np
import pytest
import torch

from ccvpe_amd import _lib, models, weights
from oracle import ccvpe_oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,crop", [(2, 320, 640, 640), (3, 320, 640, 192), (1, 154, 231, 231), (2, 16, 20, 7)])
def test_preprocess_is_bit_identical_to_the_torchvision_semantics(B, H, W, crop):
    g = torch.Generator().manual_seed(B * 1000 + W)
    img = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    shift = [int(v) for v in torch.randint(-2 * W, 2 * W, (B,), generator=g)]
    ref = orc.preprocess(img, shift, crop)
    out = _lib.preprocess(img.cuda(), shift, crop)
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert torch.equal(out.cpu(), ref), (out.cpu() - ref).abs().max().item()
    # no roll / no crop
    assert torch.equal(_lib.preprocess(img.cuda()).cpu(), orc.preprocess(img))


def test_preprocess_feeds_forward():
    cfg = gu.CONFIGS["oxford"]
    m = models.CVM_OxfordRobotCar("cuda")
    m.load_state_dict(weights.generate_state_dict("oxford", 0))
    m.to("cuda").eval()
    g = torch.Generator().manual_seed(5)
    grd_u8 = torch.randint(0, 256, (1, 154, 231, 3), dtype=torch.uint8, generator=g)
    sat_u8 = torch.randint(0, 256, (1, 512, 512, 3), dtype=torch.uint8, generator=g)
    outs = m(_lib.preprocess(grd_u8.cuda()), _lib.preprocess(sat_u8.cuda()))
    ref = m(orc.preprocess(grd_u8).cuda(), orc.preprocess(sat_u8).cuda())
    assert torch.equal(outs[0], ref[0])


@pytest.mark.parametrize("name,batch", [("oxford", 1), ("vigor_prior180_circ", 2)])
def test_cached_aerial_forward_equals_full_forward(name, batch):
    cfg = gu.CONFIGS[name]
    v = cfg["variant"]
    if v == "oxford":
        m = models.CVM_OxfordRobotCar("cuda")
    else:
        m = models.CVM_VIGOR_ori_prior("cuda", cfg["ori_noise"], cfg["circular"])
    m.load_state_dict(weights.generate_state_dict(v, cfg["seed"]))
    m.to("cuda").eval()
    g, s = weights.generate_inputs(v, batch, 7, cfg["fov"])
    g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
    full = m(g, s)
    cache = m.encode_aerial(s)
    cached = m.forward_cached(g, cache)
    # the cached plan's ground encoder / matching / decoder launches are the full plan's (same tuning-table entries) and the cached
    # taps are copies of what the full plan computes: every output, the orientation field included, has the same bits
    for i, (a, b) in enumerate(zip(full, cached)):
        assert torch.equal(a, b), gu.OUTPUT_NAMES[i]
    # a second ground frame against the same cached tile (the streaming use case)
    g2 = torch.roll(g, 37, dims=3)
    a = m(g2, s)
    b = m.forward_cached(g2, cache)
    assert torch.equal(a[0], b[0])
    with pytest.raises(ValueError):
        m.forward_cached(g[:1].repeat(3, 1, 1, 1), cache)


def test_postprocess_argmax_ties_take_the_first_index_like_numpy():
    """train_VIGOR.py:300 uses np.argmax on the flattened heat map -> first maximal index on ties."""
    m = models.CVM_OxfordRobotCar("cuda")
    m.load_state_dict(weights.generate_state_dict("oxford", 0))
    m.to("cuda").eval()
    rng = np.random.default_rng(5)
    B, n = 4, 512 * 512
    heat = rng.random((B, 1, 512, 512), dtype=np.float32) * 0.5
    flat = heat.reshape(B, n)
    flat[0, [n - 1, 77777, 4099]] = 0.9          # three equal maxima, scattered over different threads
    flat[1, [5, 4, 262143]] = 0.75               # neighbours inside one 16-byte load
    flat[2, 0] = 0.99                            # very first element
    flat[3, n - 1] = 0.99                        # very last element
    ang = rng.uniform(-np.pi, np.pi, size=(B, 1, 512, 512)).astype(np.float32)
    ori = np.concatenate([np.cos(ang), np.sin(ang)], axis=1)
    post = m.postprocess(torch.from_numpy(heat).cuda(), torch.from_numpy(ori).cuda())
    want = orc.postprocess(torch.from_numpy(heat), torch.from_numpy(ori))
    idx = post["index"].cpu().numpy()
    np.testing.assert_array_equal(idx, flat.argmax(axis=1))
    np.testing.assert_array_equal(idx, [4099, 4, 0, n - 1])
    np.testing.assert_array_equal(post["prob"].cpu().numpy(), flat.max(axis=1))
    np.testing.assert_allclose(post["angle_deg"].cpu().numpy(), want[4].numpy(), rtol=1e-5, atol=1e-3)
    # the same five numbers as float rows (ccvpe_postprocess_rows: what the data-parallel gather moves), a larger batch than the scratch was
    # sized for, then the small one again (the ticket counters of the scratch do not move), and the same bits on every repeat
    ht, ot = torch.from_numpy(heat).cuda(), torch.from_numpy(ori).cuda()
    rows = m.postprocess_rows(ht, ot)
    for k, name in enumerate(("prob", "cos", "sin", "angle_deg")):
        assert torch.equal(rows[:, k + 1], post[name]), name
    np.testing.assert_array_equal(rows[:, 0].cpu().numpy(), idx.astype(np.float32))
    big = m.postprocess_rows(ht.repeat(12, 1, 1, 1), ot.repeat(12, 1, 1, 1))
    assert torch.equal(big, rows.repeat(12, 1))
    for _ in range(3):
        assert torch.equal(m.postprocess_rows(ht, ot), rows)


def _resize_reference(img_u8, oh, ow, shift, crop_w):
    """oracle: Pillow-exact resize (numpy) then the ToTensor / Normalize / roll / crop restatement."""
    from oracle import ccvpe_oracle as orc
    from oracle import resize_oracle as ro
    rs = np.stack([ro.resize_bilinear_u8(im, oh, ow) for im in img_u8])
    return orc.preprocess(torch.from_numpy(rs), shift, crop_w)


def test_resize_preprocess_matches_pillow_fixture_bit_for_bit():
    """resize inside the pre-processing kernels.  Expected bytes come from Pillow itself
    (tests/golden/resize.npz); the float stage must equal torchvision's (x/255 - mean)/std exactly."""
    from ccvpe_amd import _lib
    from oracle import ccvpe_oracle as orc
    fx = np.load(gu.GOLDEN_DIR + "/resize.npz", allow_pickle=False)
    i = 0
    while f"in{i}" in fx:
        img, want = fx[f"in{i}"], fx[f"out{i}"]
        got = _lib.preprocess_resize(torch.from_numpy(img[None]).cuda(), want.shape[:2])
        ref = orc.preprocess(torch.from_numpy(want[None]))
        assert got.shape == ref.shape
        assert torch.equal(got.cpu(), ref), f"case {i}: {img.shape} -> {want.shape}"
        i += 1
    assert i >= 7


def test_resize_preprocess_full_vigor_geometry_with_roll_and_crop():
    """1024 x 2048 panoramas -> 320 x 640 with a per-sample roll and the HFoV-108 crop, and 640 x 640 aerial tiles -> 512 x 512:
    bit-identical to the oracle (numpy restatement of Pillow, pinned by tests/test_resize_oracle.py)."""
    from ccvpe_amd import _lib
    rng = np.random.default_rng(3)
    pano = rng.integers(0, 256, size=(2, 1024, 2048, 3), dtype=np.uint8)
    shift = [197, -45]
    got = _lib.preprocess_resize(torch.from_numpy(pano).cuda(), (320, 640), shift=shift, crop_w=192)
    assert torch.equal(got.cpu(), _resize_reference(pano, 320, 640, shift, 192))
    sat = rng.integers(0, 256, size=(2, 640, 640, 3), dtype=np.uint8)
    got = _lib.preprocess_resize(torch.from_numpy(sat).cuda(), (512, 512))
    assert torch.equal(got.cpu(), _resize_reference(sat, 512, 512, None, None))
    # an axis that keeps its size is not resampled (Pillow skips the pass): width-only and height-only
    img = rng.integers(0, 256, size=(1, 96, 200, 3), dtype=np.uint8)
    assert torch.equal(_lib.preprocess_resize(torch.from_numpy(img).cuda(), (96, 64)).cpu(), _resize_reference(img, 96, 64, None, None))
    assert torch.equal(_lib.preprocess_resize(torch.from_numpy(img).cuda(), (40, 200)).cpu(), _resize_reference(img, 40, 200, None, None))
    with pytest.raises(_lib.CcvpeError):
        _lib.preprocess_resize(torch.from_numpy(img).cuda(), (8, 8))       # 12x / 25x down-scaling: refused, not truncated


def test_ground_truth_side_metrics_match_the_restated_test_loop():
    """SURVEY 8f row 1, second half: pixel -> metre distance, probability at the GT pixel, orientation error and the KITTI
    lateral / longitudinal split on device, against the line-by-line numpy restatement of train_VIGOR.py:296-326 /
    train_KITTI.py:318-325 (oracle/ccvpe_oracle.py eval_metrics)."""
    cfg = gu.CONFIGS["kitti"]
    m = models.CVM_KITTI("cuda")
    m.load_state_dict(weights.generate_state_dict("kitti", 0))
    m.to("cuda").eval()
    B = 6
    g, s = weights.generate_inputs("kitti", B, 4)
    outs = m(torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda())
    heat, ori = outs[1], outs[2]
    rng = np.random.default_rng(0)
    gt_index = rng.integers(0, 512 * 512, size=B)
    gt_index[0] = int(heat[0].flatten().argmax())          # a perfect hit: distance 0, atan2(0, 0)
    ang = rng.uniform(0, 2 * np.pi, size=B)
    gt_cs = np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)
    mpp = rng.uniform(0.1, 0.3, size=B)
    heading = rng.uniform(-180, 180, size=B)
    got = m.evaluate(heat, ori, gt_index, mpp, gt_cs, heading)
    ref = orc.eval_metrics(heat.cpu().numpy(), ori.cpu().numpy(), gt_index, mpp, gt_cs, heading)
    for k, v in ref.items():
        gk = got[k].cpu().numpy()
        assert np.array_equal(np.isnan(gk), np.isnan(v)), k
        assert np.allclose(gk, v, rtol=1e-9, atol=1e-7, equal_nan=True), (k, gk, v)
    # VIGOR form: no heading -> no lateral / longitudinal numbers; scalar metres-per-pixel broadcast
    got = m.evaluate(heat, ori, gt_index, 0.113248 / 512 * 640, gt_cs)
    ref = orc.eval_metrics(heat.cpu().numpy(), ori.cpu().numpy(), gt_index, 0.113248 / 512 * 640, gt_cs)
    assert torch.isnan(got["lateral_m"]).all()
    assert np.allclose(got["meter_distance"].cpu().numpy(), ref["meter_distance"], rtol=1e-12)


def test_packed_weight_cache_round_trip(tmp_path):
    """the folded / repacked / Winograd-transformed device weights are written once and loaded by a later
    handle instead of re-packing the checkpoint; outputs are bit-identical, a different state dict gets its own file, and a
    corrupt file falls back to the state dict."""
    import os
    cfg = gu.CONFIGS["oxford"]
    sd = weights.generate_state_dict("oxford", 5)
    g, s = weights.generate_inputs("oxford", 1, 5)
    g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()

    def make(sdict):
        m = models.CVM_OxfordRobotCar("cuda", weight_cache=str(tmp_path))
        m.load_state_dict(sdict)
        return m.to("cuda").eval()

    a = make(sd)
    ref = [o.clone() for o in a(g, s)]
    assert a.last_weight_source == "state_dict"
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 1 and files[0].endswith(".ccvpepack")
    b = make(sd)
    outs = b(g, s)
    assert b.last_weight_source == "packed-cache"
    # same packed bits and the same tuning table (the session cache: tests/conftest.py): bit-identical outputs
    for i, (x, y) in enumerate(zip(ref, outs)):
        assert torch.equal(x, y), gu.OUTPUT_NAMES[i]
    c = make(weights.generate_state_dict("oxford", 6))          # other weights -> other key, not the cached file
    c(g, s)
    assert c.last_weight_source == "state_dict" and len(os.listdir(tmp_path)) == 2
    with open(os.path.join(tmp_path, files[0]), "r+b") as fh:   # truncate: the loader must refuse, the model re-packs
        fh.truncate(4096)
    d = make(sd)
    outs = d(g, s)
    assert d.last_weight_source == "state_dict"
    for i, (x, y) in enumerate(zip(ref, outs)):
        assert torch.equal(x, y), gu.OUTPUT_NAMES[i]


def test_workspace_bytes_of_the_headline_plan_is_bounded():
    """ccvpe_workspace_bytes: the post-reuse arena of the batch-32 VIGOR plan (3.84 GB incl. two 128 MiB split-K slabs;
    13 GB before lifetime-based reuse) and its growth with the batch."""
    import ctypes as C
    from ccvpe_amd import _lib
    cfg = gu.CONFIGS["vigor_prior180_circ"]
    m = models.CVM_VIGOR_ori_prior("cuda", cfg["ori_noise"], cfg["circular"])
    m.load_state_dict(weights.generate_state_dict(cfg["variant"], cfg["seed"]))
    m.to("cuda").eval()
    m._ensure_handle(torch.device("cuda", torch.cuda.current_device()))
    lib = _lib.load()
    sizes = {b: lib.ccvpe_workspace_bytes(m._handle, b, 320, 640) for b in (1, 8, 32)}
    assert 0 < sizes[1] < sizes[8] < sizes[32] < 6 * 1024 ** 3, sizes
    assert sizes[32] > 2 * 1024 ** 3, sizes       # 32 samples x the 256 x 256 x 96 expanded tensor alone are 0.8 GB
    assert lib.ccvpe_workspace_bytes(m._handle, 0, 320, 640) == 0      # bad batch: 0 and an error message
