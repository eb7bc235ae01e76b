"""Generate tests/golden/resize.npz with Pillow itself (build container only; Pillow is what torchvision's
transforms.Resize calls for PIL images, train_VIGOR.py:57-70).  Data only: uint8 inputs and PIL's uint8 outputs.

    python -m oracle.make_resize_golden
"""
import os

import numpy as np
import PIL
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resize.npz")

# (in_h, in_w) -> (out_h, out_w): the reference's ratios at reduced size (VIGOR panorama 1024x2048 -> 320x640 = 3.2x down,
# aerial 640x640 -> 512x512 = 1.25x down), KITTI-like anisotropic, an upscale, an identity axis and odd sizes
CASES = [((205, 410), (64, 128)), ((80, 80), (64, 64)), ((94, 311), (64, 256)), ((40, 60), (64, 96)), ((64, 100), (64, 37)),
         ((77, 115), (33, 49)), ((7, 5), (3, 9))]


def main():
    rng = np.random.default_rng(2024)
    data = {"pillow_version": np.array(PIL.__version__)}
    for i, ((ih, iw), (oh, ow)) in enumerate(CASES):
        if i % 2 == 0:
            img = rng.integers(0, 256, size=(ih, iw, 3), dtype=np.uint8)
        else:   # smooth image with saturated regions (exercises clip8 and rounding ties less randomly)
            yy, xx = np.mgrid[0:ih, 0:iw]
            img = np.stack([(np.sin(xx / 3.0) * 140 + 128), (yy * 255.0 / max(ih - 1, 1)), ((xx + yy) % 2) * 255.0], axis=-1)
            img = np.clip(img, 0, 255).astype(np.uint8)
        out = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))
        data[f"in{i}"] = img
        data[f"out{i}"] = out
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
