"""Derives tests/golden/example_output_mask.npz from the reference's only image, /root/reference/example_output.png (README.md:9): a screenshot of the
minifb window (802 x 839: a one-pixel frame and a 38-pixel title bar around the 800 x 800 canvas) showing an OLDER version of the teapot scene (no mirror,
untextured teapot).  What is kept is a fixture, not the image: the 800 x 800 boolean mask "this canvas pixel is not the miss colour (white)", packed to bits.
It pins nothing numerically -- only camera, orientation (y up, no vertical flip, row 0 = top) and the white-miss convention, which tests/test_oracle_kat.py
checks by comparing the mask with the oracle's frame of model2.obj at 800 x 800 outside the region where the scenes differ (the mirror).

    python tests/golden/make_example_mask.py        (needs /root/reference; PIL)"""
import os
import numpy as np
from PIL import Image
HERE = os.path.dirname(os.path.abspath(__file__))
im = np.asarray(Image.open("/root/reference/example_output.png").convert("RGB"))
assert im.shape == (839, 802, 3)
canvas = im[38:838, 1:801]                                       # title bar 38 px, frame 1 px
mask = ~((canvas[..., 0] >= 250) & (canvas[..., 1] >= 250) & (canvas[..., 2] >= 250))
np.savez_compressed(os.path.join(HERE, "example_output_mask.npz"), mask_bits=np.packbits(mask), shape=np.array(mask.shape),
                    note="non-white mask of the 800x800 canvas of the reference's example_output.png (older scene: no mirror)")
print("mask:", mask.shape, "covered fraction", mask.mean(), "rows with content", np.flatnonzero(mask.any(1))[[0, -1]], "cols", np.flatnonzero(mask.any(0))[[0, -1]])
