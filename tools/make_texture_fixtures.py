"""One-off: decode the reference's JPEG textures to raw RGB8 and store them compressed.

The reference decodes with the Rust `image` crate (texture.rs:91-92); that decoder is not available here, so
the texel bytes are pinned by decoding once with Pillow in the build container. Both the product and the
oracle read these bytes, which takes the JPEG decoder out of the parity loop.

    python tools/make_texture_fixtures.py /root/reference/raytracer
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

src = sys.argv[1]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "data")
os.makedirs(out, exist_ok=True)
for name in ("earthmap.jpg", "star1.jpg", "star3.jpeg", "star5.jpeg", "star6.jpeg"):
    img = np.asarray(Image.open(os.path.join(src, name)).convert("RGB"), dtype=np.uint8)
    stem = name.split(".")[0]
    np.savez_compressed(os.path.join(out, stem + "_rgb8.npz"), rgb=img)
    print(name, img.shape, hashlib.sha256(img.tobytes()).hexdigest()[:16])
