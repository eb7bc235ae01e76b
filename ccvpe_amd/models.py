"""Host-side mirror of the reference `models` module for the inference path.

Exports the four classes the reference drivers import (train_VIGOR.py:17-18, train_KITTI.py:17,
train_OxfordRobotCar.py:17) with the same constructor arguments, the same 818-key `state_dict`
layout (strict `load_state_dict` of a reference checkpoint succeeds) and the same
`forward(grd, sat) -> 9-tuple` contract (models.py:150, 448, 752, 1051).  The modules own ordinary
torch parameters (so `.to()`, `.eval()`, `.state_dict()` behave), but `forward` does not run a
PyTorch graph: it hands device pointers to libccvpe_hip.so through the C ABI in include/ccvpe.h.

Inference only: no autograd through the HIP path, `eval()` mode required, GPU required.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, spec, tuning

__all__ = ["CVM_VIGOR", "CVM_VIGOR_ori_prior", "CVM_KITTI", "CVM_OxfordRobotCar"]


# ---------------------------------------------------------------------------------------------
# parameter containers (no forward of their own): they only reproduce the reference key layout
# ---------------------------------------------------------------------------------------------
class _Params(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the forward pass runs in libccvpe_hip.so")


def _bn(c: int) -> nn.BatchNorm2d:
    # momentum = 1 - 0.99, eps 1e-3 (efficientnet_pytorch/model.py:171-172, utils.py:665-666)
    return nn.BatchNorm2d(c, momentum=0.01, eps=spec.BN_EPS)


class _MBConvParams(_Params):
    """Key layout of MBConvBlock (efficientnet_pytorch/model.py:57-87)."""

    def __init__(self, e: int, k: int, s: int, cin: int, cout: int):
        super().__init__()
        mid = cin * e
        if e != 1:
            self._expand_conv = nn.Conv2d(cin, mid, 1, bias=False)
            self._bn0 = _bn(mid)
        self._depthwise_conv = nn.Conv2d(mid, mid, k, stride=s, groups=mid, bias=False)
        self._bn1 = _bn(mid)
        sq = spec.se_squeeze(cin)
        self._se_reduce = nn.Conv2d(mid, sq, 1)
        self._se_expand = nn.Conv2d(sq, mid, 1)
        self._project_conv = nn.Conv2d(mid, cout, 1, bias=False)
        self._bn2 = _bn(cout)


class _EfficientNetParams(_Params):
    """Key layout of EfficientNet-B0 as CCVPE builds it (efficientnet_pytorch/model.py:162-219)."""

    def __init__(self):
        super().__init__()
        self._conv_stem = nn.Conv2d(3, spec.STEM_CH, 3, stride=2, bias=False)
        self._bn0 = _bn(spec.STEM_CH)
        self._blocks = nn.ModuleList([_MBConvParams(*b) for b in spec.B0_BLOCKS])
        self._conv_head = nn.Conv2d(spec.B0_BLOCKS[-1][4], spec.HEAD_CH, 1, bias=False)
        self._bn1 = _bn(spec.HEAD_CH)
        self._fc = nn.Linear(spec.HEAD_CH, spec.FC_CLASSES)   # unused by CCVPE, present in checkpoints


def _double_conv(cin: int, mid: int, cout: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, mid, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(mid, cout, 3, padding=1))


class _CVMBase(nn.Module):
    _variant: str = ""

    def __init__(self, device, circular_padding: bool = False, ori_noise: Optional[float] = None,
                 micro_batch: int = 0, precision: str = "fp32", weight_cache: Optional[str] = None):
        super().__init__()
        v = spec.VARIANTS[self._variant]
        self.device = device
        self.circular_padding = bool(circular_padding)
        self.ori_noise = ori_noise
        self._micro_batch = int(micro_batch)
        if precision not in ("fp32", "bf16x3"):
            raise ValueError("precision must be 'fp32' (exact fp32 MFMA) or 'bf16x3' (3-term bf16 split, ~1e-5 relative)")
        self._precision = precision
        # packed-weight cache directory: None = environment CCVPE_WEIGHT_CACHE, empty = off
        self._weight_cache = weight_cache if weight_cache is not None else os.environ.get("CCVPE_WEIGHT_CACHE", "")
        self.last_weight_source = None   # "packed-cache" or "state_dict" after the first forward (introspection / tests)

        self.grd_efficientnet = _EfficientNetParams()
        for lvl, c in enumerate(v.head_ch, 1):
            setattr(self, f"grd_feature_to_descriptor{lvl}", nn.Sequential(
                nn.Conv2d(spec.HEAD_CH, c, 1), nn.Identity(), nn.Conv2d(v.feat_h, 1, 1), nn.Flatten(start_dim=1)))
        self.sat_efficientnet = _EfficientNetParams()
        self.sat_feature_to_descriptors = nn.Sequential(nn.Flatten(start_dim=1), nn.Linear(spec.HEAD_CH * 4, v.sat_desc))
        for sfx, dec in (("", v.loc), ("_ori", v.ori)):
            for j, lv in enumerate(dec):
                n = 6 - j
                setattr(self, f"deconv{n}{sfx}", nn.ConvTranspose2d(lv.deconv_in, lv.deconv_out, 2, 2))
                setattr(self, f"conv{n}{sfx}", _double_conv(lv.deconv_out + lv.skip, lv.mid, lv.out))

        self._handle: Optional[C.c_void_p] = None
        self._handle_device: Optional[int] = None
        self._weights_dirty = True
        self._debug = False
        self._rolls: Tuple[int, ...] = ()

    # ---- lifetime of the native handle ------------------------------------------------------
    def _release(self):
        if getattr(self, "_handle", None) is not None:
            try:
                _lib.load().ccvpe_destroy(self._handle)
            finally:
                self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _apply(self, fn, *a, **k):       # .to() / .cuda() / .cpu() move the parameters
        out = super()._apply(fn, *a, **k)
        self._weights_dirty = True
        return out

    def load_state_dict(self, state_dict, strict: bool = True, *a, **k):
        out = super().load_state_dict(state_dict, strict, *a, **k)
        self._weights_dirty = True
        return out

    def refresh_weights(self) -> None:
        """Call after modifying parameters in place: the next forward re-ingests the state_dict."""
        self._weights_dirty = True

    def _ensure_handle(self, dev: torch.device) -> None:
        lib = _lib.load()
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._handle is not None and self._handle_device != index:
            self._release()
        if self._handle is None:
            cfg = _lib.Config()
            cfg.variant = _lib.VARIANT_ID[self._variant]
            cfg.circular_padding = int(self.circular_padding)
            cfg.ori_noise = float(self.ori_noise) if self.ori_noise is not None else 0.0
            cfg.device = index
            cfg.micro_batch = self._micro_batch
            cfg.reserved[0] = 1 if self._precision == "bf16x3" else 0
            h = C.c_void_p()
            _lib.check(lib.ccvpe_create(C.byref(cfg), C.byref(h)), "ccvpe_create")
            self._handle, self._handle_device = h, index
            self._weights_dirty = True
            self._rolls = tuple(lib.ccvpe_output_channels(h, k) for k in range(6))
            tuning.load_into(lib, h)     # committed table + this machine's cache: known plans are not measured again
            self._tune_gen = lib.ccvpe_tuning_generation(h)
            if self._debug:
                _lib.check(lib.ccvpe_set_debug(h, 1), "ccvpe_set_debug")
            if getattr(self, "_n_streams", 2) != 2:
                _lib.check(lib.ccvpe_set_streams(h, self._n_streams), "ccvpe_set_streams")
        if self._weights_dirty:
            sd = self.state_dict()
            cache_file = self._cache_path(sd) if self._weight_cache else None
            if cache_file and os.path.exists(cache_file) and lib.ccvpe_load_packed(self._handle, cache_file.encode()) == 0:
                self.last_weight_source = "packed-cache"
            else:
                for key, t in sd.items():
                    if t.dtype != torch.float32 or "._fc." in key:
                        _lib.check(lib.ccvpe_skip_weight(self._handle, key.encode()), f"ccvpe_skip_weight({key})")
                        continue
                    t = t.detach().contiguous()
                    shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                    _lib.check(lib.ccvpe_set_weight(self._handle, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()),
                               f"ccvpe_set_weight({key})")
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                _lib.check(lib.ccvpe_finalize_weights(self._handle), "ccvpe_finalize_weights")
                self.last_weight_source = "state_dict"
                if cache_file:   # best effort: a lost race or a read-only directory costs the re-pack on the next start only
                    try:
                        os.makedirs(os.path.dirname(cache_file), exist_ok=True)
                        if lib.ccvpe_save_packed(self._handle, cache_file.encode()) != 0:
                            import warnings
                            warnings.warn(f"ccvpe_amd: packed-weight cache not written: {(lib.ccvpe_last_error() or b'').decode()}")
                    except OSError as e:
                        import warnings
                        warnings.warn(f"ccvpe_amd: packed-weight cache not written: {e}")
            self._weights_dirty = False

    def _tuning_sync(self) -> None:
        """After a call that may have built a plan: write the tuning table back when this handle has measured a new one."""
        gen = _lib.load().ccvpe_tuning_generation(self._handle)
        if gen != getattr(self, "_tune_gen", 0):
            self._tune_gen = gen
            tuning.save_from(_lib.load(), self._handle)

    def export_tuning(self) -> str:
        """Text of this handle's tuning table (plans loaded at start plus plans measured since)."""
        return tuning.export(_lib.load(), self._handle) if self._handle is not None else ""

    def _cache_path(self, sd) -> str:
        """File name of the packed weights of this exact state dict: content hash of every float tensor in key order,
        variant, precision, padding mode and the digest of the library sources (a rebuilt library repacks)."""
        import hashlib
        try:
            import xxhash
            hsh = xxhash.xxh3_128()
        except ImportError:
            hsh = hashlib.sha256()
        for key in sorted(sd):
            t = sd[key]
            if t.dtype != torch.float32 or "._fc." in key:
                continue
            hsh.update(key.encode())
            hsh.update(t.detach().cpu().contiguous().numpy().tobytes())
        # switches that change what the packer emits are part of the key, and so is the library actually loaded
        sw = _lib.load().ccvpe_pack_switches() or b""   # the library's own list: whatever its packer branches on
        if sw:
            hsh.update(sw)
        tag = f"{self._variant}-{self._precision}-{int(self.circular_padding)}-{_lib.library_digest()[:16]}-{hsh.hexdigest()[:32]}"
        return os.path.join(self._weight_cache, tag + ".ccvpepack")

    # ---- forward ----------------------------------------------------------------------------
    def _prepare(self, grd: torch.Tensor, sat: torch.Tensor):
        if self.training:
            raise RuntimeError("ccvpe_amd runs the inference path only: call .eval() first "
                               "(reference test loops do, train_VIGOR.py:254)")
        if not (grd.is_cuda and sat.is_cuda):
            raise RuntimeError("ccvpe_amd has no CPU path: inputs must live on an MI355X (cuda) device")
        if grd.device != sat.device:
            raise RuntimeError("grd and sat must be on the same device")
        if grd.dim() != 4 or sat.dim() != 4 or grd.shape[1] != 3 or sat.shape[1] != 3 or grd.shape[0] != sat.shape[0]:
            raise ValueError(f"expected grd [B,3,H,W] and sat [B,3,512,512], got {tuple(grd.shape)} / {tuple(sat.shape)}")
        if tuple(sat.shape[2:]) != spec.SAT_HW:
            raise ValueError(f"aerial input must be 3x512x512, got {tuple(sat.shape)}")
        grd = grd.detach().to(torch.float32).contiguous()
        sat = sat.detach().to(torch.float32).contiguous()
        self._ensure_handle(grd.device)
        return grd, sat

    def _alloc_outputs(self, B: int, dev: torch.device):
        n = spec.OUT_HW[0] * spec.OUT_HW[1]
        logits = torch.empty((B, n), dtype=torch.float32, device=dev)
        heat = torch.empty((B, 1) + spec.OUT_HW, dtype=torch.float32, device=dev)
        ori = torch.empty((B, 2) + spec.OUT_HW, dtype=torch.float32, device=dev)
        ms = [torch.empty((B, self._rolls[k], 8 << k, 8 << k), dtype=torch.float32, device=dev) for k in range(6)]
        out = _lib.Outputs()
        out.logits_flattened = logits.data_ptr()
        out.heatmap = heat.data_ptr()
        out.ori = ori.data_ptr()
        for k in range(6):
            out.matching_score[k] = ms[k].data_ptr()
        return out, (logits, heat, ori, *ms)

    def forward(self, grd: torch.Tensor, sat: torch.Tensor):
        grd, sat = self._prepare(grd, sat)
        B = grd.shape[0]
        with torch.cuda.device(grd.device):
            out, tensors = self._alloc_outputs(B, grd.device)
            stream = torch.cuda.current_stream(grd.device).cuda_stream
            rc = _lib.load().ccvpe_forward(self._handle, C.c_void_p(grd.data_ptr()), grd.shape[2], grd.shape[3],
                                           C.c_void_p(sat.data_ptr()), B, C.byref(out), C.c_void_p(stream))
        _lib.check(rc, "ccvpe_forward")
        self._tuning_sync()
        return tensors

    # ---- aerial-side caching for streaming -------------------------------
    def encode_aerial(self, sat: torch.Tensor) -> torch.Tensor:
        """Encode aerial images once; the returned device buffer feeds forward_cached() for any number of
        ground frames that share the tile (Oxford RobotCar reuses tiles, datasets.py:306-317)."""
        if self.training:
            raise RuntimeError("ccvpe_amd runs the inference path only: call .eval() first")
        if not sat.is_cuda or sat.dim() != 4 or tuple(sat.shape[1:]) != (3,) + spec.SAT_HW:
            raise ValueError("sat must be a cuda tensor [B,3,512,512]")
        sat = sat.detach().to(torch.float32).contiguous()
        self._ensure_handle(sat.device)
        lib = _lib.load()
        B = sat.shape[0]
        nbytes = lib.ccvpe_aerial_cache_bytes(self._handle, B)
        cache = torch.empty(nbytes // 4, dtype=torch.float32, device=sat.device)
        stream = torch.cuda.current_stream(sat.device).cuda_stream
        _lib.check(lib.ccvpe_encode_aerial(self._handle, C.c_void_p(sat.data_ptr()), B, C.c_void_p(cache.data_ptr()),
                                           C.c_void_p(stream)), "ccvpe_encode_aerial")
        self._tuning_sync()
        cache._ccvpe_batch = B
        return cache

    def forward_cached(self, grd: torch.Tensor, cache: torch.Tensor):
        """forward(grd, sat) with the aerial side taken from encode_aerial(sat)."""
        if self.training:
            raise RuntimeError("ccvpe_amd runs the inference path only: call .eval() first")
        if not grd.is_cuda or grd.dim() != 4 or grd.shape[1] != 3:
            raise ValueError("grd must be a cuda tensor [B,3,H,W]")
        grd = grd.detach().to(torch.float32).contiguous()
        self._ensure_handle(grd.device)
        B = grd.shape[0]
        if getattr(cache, "_ccvpe_batch", B) != B:
            raise ValueError("cache was encoded for a different batch size")
        out, tensors = self._alloc_outputs(B, grd.device)
        stream = torch.cuda.current_stream(grd.device).cuda_stream
        rc = _lib.load().ccvpe_forward_cached(self._handle, C.c_void_p(grd.data_ptr()), grd.shape[2], grd.shape[3],
                                              C.c_void_p(cache.data_ptr()), B, C.byref(out), C.c_void_p(stream))
        _lib.check(rc, "ccvpe_forward_cached")
        self._tuning_sync()
        return tensors

    # ---- extras beyond the reference surface ------------------------------------------------
    def postprocess(self, heatmap: torch.Tensor, ori: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Device-side version of the per-sample loop in train_VIGOR.py:297-316."""
        B = heatmap.shape[0]
        self._ensure_handle(heatmap.device)
        buf = torch.empty((B, 5), dtype=torch.int32, device=heatmap.device)
        stream = torch.cuda.current_stream(heatmap.device).cuda_stream
        rc = _lib.load().ccvpe_postprocess(self._handle, C.c_void_p(heatmap.contiguous().data_ptr()),
                                           C.c_void_p(ori.contiguous().data_ptr()), B, C.c_void_p(buf.data_ptr()),
                                           C.c_void_p(stream))
        _lib.check(rc, "ccvpe_postprocess")
        f = buf.view(torch.float32)
        return {"index": buf[:, 0].to(torch.int64), "prob": f[:, 1], "cos": f[:, 2], "sin": f[:, 3], "angle_deg": f[:, 4]}

    def postprocess_rows(self, heatmap: torch.Tensor, ori: torch.Tensor) -> torch.Tensor:
        """postprocess() as ONE float32 tensor [B, 5] = (index, prob, cos, sin, angle_deg): the 20-byte rows a data-parallel
        evaluation gathers, written by the post-processing launch itself (no conversion / stack launches behind it)."""
        B = heatmap.shape[0]
        self._ensure_handle(heatmap.device)
        rows = torch.empty((B, 5), dtype=torch.float32, device=heatmap.device)
        stream = torch.cuda.current_stream(heatmap.device).cuda_stream
        rc = _lib.load().ccvpe_postprocess_rows(self._handle, C.c_void_p(heatmap.contiguous().data_ptr()),
                                                C.c_void_p(ori.contiguous().data_ptr()), B, C.c_void_p(rows.data_ptr()),
                                                C.c_void_p(stream))
        _lib.check(rc, "ccvpe_postprocess_rows")
        return rows

    METRIC_FIELDS = ("pixel_distance", "meter_distance", "prob_at_gt", "angle_pred_deg", "angle_gt_deg", "orientation_error_deg",
                     "longitudinal_m", "lateral_m")

    def evaluate(self, heatmap: torch.Tensor, ori: torch.Tensor, gt_index, meter_per_pixel, gt_cos_sin=None, heading_deg=None
                 ) -> Dict[str, torch.Tensor]:
        """Device-side version of the whole per-sample test loop (train_VIGOR.py:290-326, train_KITTI.py:309-343): argmax /
        orientation lookup (postprocess) plus the ground-truth side - pixel and metre distance, probability at the
        ground-truth pixel, orientation error, lateral / longitudinal split.  Returns float64 tensors [B] (NaN where the
        reference produces no value for the query)."""
        B = heatmap.shape[0]
        dev = heatmap.device
        self._ensure_handle(dev)
        heatmap = heatmap.contiguous()
        pose = torch.empty((B, 5), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        lib = _lib.load()
        _lib.check(lib.ccvpe_postprocess(self._handle, C.c_void_p(heatmap.data_ptr()), C.c_void_p(ori.contiguous().data_ptr()), B,
                                         C.c_void_p(pose.data_ptr()), C.c_void_p(stream)), "ccvpe_postprocess")
        gi = torch.as_tensor(gt_index, dtype=torch.int32, device=dev).contiguous()
        mpp = torch.as_tensor(meter_per_pixel, dtype=torch.float64, device=dev).expand(B).contiguous()
        gcs = None if gt_cos_sin is None else torch.as_tensor(gt_cos_sin, dtype=torch.float32, device=dev).reshape(B, 2).contiguous()
        hd = None if heading_deg is None else torch.as_tensor(heading_deg, dtype=torch.float64, device=dev).expand(B).contiguous()
        out = torch.empty((B, len(self.METRIC_FIELDS)), dtype=torch.float64, device=dev)
        _lib.check(lib.ccvpe_eval_metrics(self._handle, C.c_void_p(pose.data_ptr()), C.c_void_p(heatmap.data_ptr()), B, C.c_void_p(gi.data_ptr()),
                                          C.c_void_p(gcs.data_ptr()) if gcs is not None else None, C.c_void_p(mpp.data_ptr()),
                                          C.c_void_p(hd.data_ptr()) if hd is not None else None, C.c_void_p(out.data_ptr()), C.c_void_p(stream)),
                   "ccvpe_eval_metrics")
        res = {k: out[:, i] for i, k in enumerate(self.METRIC_FIELDS)}
        res["index"] = pose[:, 0].to(torch.int64)
        return res

    def set_streams(self, n: int) -> None:
        """Issue order of later forwards: 2 (default) two-stream schedule, 1 program order (bit-identical results)."""
        self._n_streams = int(n)
        if self._handle is not None:
            _lib.check(_lib.load().ccvpe_set_streams(self._handle, int(n)), "ccvpe_set_streams")

    def set_debug(self, enable: bool) -> None:
        self._debug = bool(enable)
        if self._handle is not None:
            _lib.check(_lib.load().ccvpe_set_debug(self._handle, int(enable)), "ccvpe_set_debug")

    def read_tap(self, name: str) -> torch.Tensor:
        """Intermediate tensor of the last debug forward as NCHW float32 on the CPU."""
        lib = _lib.load()
        cap = 64 * 1024 * 1024
        while True:
            host = torch.empty(cap, dtype=torch.float32)
            n = C.c_size_t(0)
            shape = (C.c_int32 * 4)()
            rc = lib.ccvpe_read_tap(self._handle, name.encode(), C.c_void_p(host.data_ptr()), cap, C.byref(n), C.byref(shape))
            if rc == -1 and b"needs" in (lib.ccvpe_last_error() or b""):
                cap *= 4
                continue
            _lib.check(rc, f"ccvpe_read_tap({name})")
            return host[: n.value].reshape(*[int(s) for s in shape]).clone()

    def profile(self, grd: torch.Tensor, sat: torch.Tensor):
        """One forward with a hipEvent pair around every launch: list of (name, ms, flops, bytes, issued_flops)."""
        grd, sat = self._prepare(grd, sat)
        B = grd.shape[0]
        lib = _lib.load()
        out, _ = self._alloc_outputs(B, grd.device)
        stream = torch.cuda.current_stream(grd.device).cuda_stream
        n = lib.ccvpe_profile_forward(self._handle, C.c_void_p(grd.data_ptr()), grd.shape[2], grd.shape[3],
                                      C.c_void_p(sat.data_ptr()), B, C.byref(out), C.c_void_p(stream))
        _lib.check(n, "ccvpe_profile_forward")
        self._tuning_sync()
        rows = []
        name = C.create_string_buffer(128)
        ms, fl, by, iss = C.c_float(), C.c_double(), C.c_double(), C.c_double()
        for i in range(n):
            _lib.check(lib.ccvpe_profile_row(self._handle, i, name, 128, C.byref(ms), C.byref(fl), C.byref(by)), "ccvpe_profile_row")
            _lib.check(lib.ccvpe_profile_row_issued(self._handle, i, C.byref(iss)), "ccvpe_profile_row_issued")
            rows.append((name.value.decode(), ms.value, fl.value, by.value, iss.value))
        return rows


class CVM_VIGOR(_CVMBase):
    """models.py:49 - CVM_VIGOR(device, circular_padding)."""
    _variant = "vigor"

    def __init__(self, device, circular_padding, **kw):
        super().__init__(device, circular_padding, None, **kw)


class CVM_VIGOR_ori_prior(_CVMBase):
    """models.py:346 - CVM_VIGOR_ori_prior(device, ori_noise, circular_padding=True)."""
    _variant = "vigor_ori_prior"

    def __init__(self, device, ori_noise, circular_padding=True, **kw):
        super().__init__(device, circular_padding, ori_noise, **kw)


class CVM_KITTI(_CVMBase):
    """models.py:655 - CVM_KITTI(device); no circular padding (models.py:660)."""
    _variant = "kitti"

    def __init__(self, device, **kw):
        super().__init__(device, False, None, **kw)


class CVM_OxfordRobotCar(_CVMBase):
    """models.py:954 - CVM_OxfordRobotCar(device)."""
    _variant = "oxford"

    def __init__(self, device, **kw):
        super().__init__(device, False, None, **kw)
