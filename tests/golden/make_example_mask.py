"""Derives tests/golden/example_output_mask.npz from the reference's only image, /root/reference/example_output.png (README.md:9): a screenshot of the
minifb window (802 x 839: a one-pixel frame and a 38-pixel title bar around the 800 x 800 canvas) showing an OLDER version of the teapot scene (no mirror,
another table-top material; same camera, lights, teapot and table front).  What is kept is a fixture of reference-produced OUTPUT data, not the image:
  * mask_bits: the 800 x 800 boolean mask "this canvas pixel is not the miss colour (white)", packed to bits;
  * teapot_rgb / front_rgb: the canvas pixels of two rectangles -- the teapot's bounding box (rows 280..570, cols 120..660) and the table's front face
    (rows 695..750, cols 0..800) -- as the reference rendered them.
tests/test_oracle_kat.py compares the oracle's frame of model2.obj at 800 x 800 with them: silhouette everywhere outside the mirror, colours on the teapot
and the table front (the parts of the old scene that the new one still contains).

    python tests/golden/make_example_mask.py        (needs /root/reference; PIL)"""
import os
import numpy as np
from PIL import Image
HERE = os.path.dirname(os.path.abspath(__file__))
im = np.asarray(Image.open("/root/reference/example_output.png").convert("RGB"))
assert im.shape == (839, 802, 3)
canvas = im[38:838, 1:801]                                       # title bar 38 px, frame 1 px
mask = ~((canvas[..., 0] >= 250) & (canvas[..., 1] >= 250) & (canvas[..., 2] >= 250))
TEAPOT = (280, 570, 120, 660); FRONT = (695, 750, 0, 800)
np.savez_compressed(os.path.join(HERE, "example_output_mask.npz"), mask_bits=np.packbits(mask), shape=np.array(mask.shape),
                    teapot_box=np.array(TEAPOT), teapot_rgb=canvas[TEAPOT[0]:TEAPOT[1], TEAPOT[2]:TEAPOT[3]].copy(),
                    front_box=np.array(FRONT), front_rgb=canvas[FRONT[0]:FRONT[1], FRONT[2]:FRONT[3]].copy(),
                    note="derived from the 800x800 canvas of the reference's example_output.png (older scene: no mirror, other table top)")
print("mask:", mask.shape, "covered fraction", mask.mean(), "rows with content", np.flatnonzero(mask.any(1))[[0, -1]], "cols", np.flatnonzero(mask.any(0))[[0, -1]])
