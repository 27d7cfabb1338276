"""Manual check (not collected by pytest): byte-level mutations of small JPEG / PNG files through rrt_decode_image_file -- any outcome but a crash is fine.
Meant to be run against a host-sanitizer build of the library (hipcc ... -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -shared-libsan):
   RRT_LIB=/tmp/asan/librrt_hip_asan.so RRT_NO_TORCH_PRELOAD=1 LD_PRELOAD=<libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0 python tests/manual/fuzz_decoders.py [iterations]"""
import importlib, io, os, sys, tempfile
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rrt = importlib.import_module("rust-ray-tracer_amd")
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(77)
img = np.clip(np.cumsum(rng.normal(size=(41, 67, 3)), axis=1) * 9 + 128, 0, 255).astype(np.uint8)
seeds = []
for kw in (dict(quality=80, subsampling=0), dict(quality=60, subsampling=2), dict(quality=90, subsampling=1, progressive=True), dict(quality=70, subsampling=2, restart_marker_blocks=2)):
    b = io.BytesIO(); Image.fromarray(img).save(b, "JPEG", **kw); seeds.append(("jpg", b.getvalue()))
for kw in (dict(), dict(interlace=True)):
    b = io.BytesIO(); Image.fromarray(img).save(b, "PNG", **kw); seeds.append(("png", b.getvalue()))
b = io.BytesIO(); Image.fromarray(img).quantize(16).save(b, "PNG", bits=4); seeds.append(("png", b.getvalue()))
b = io.BytesIO(); Image.fromarray(img).save(b, "BMP"); seeds.append(("bmp", b.getvalue()))
b = io.BytesIO(); Image.fromarray(img).quantize(16).save(b, "BMP"); seeds.append(("bmp", b.getvalue()))
for kw in (dict(), dict(compression="tga_rle")):
    b = io.BytesIO(); Image.fromarray(img).save(b, "TGA", **kw); seeds.append(("tga", b.getvalue()))
ok = err = 0
os.environ["RRT_JPEG_PART_BYTES"] = "64"; os.environ["RRT_HOST_THREADS"] = "8"
with tempfile.TemporaryDirectory() as d:
    for it in range(n_iter):
        ext, data = seeds[it % len(seeds)]
        a = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            mode = int(rng.integers(0, 4)); pos = int(rng.integers(0, len(a)))
            if mode == 0: a[pos] = int(rng.integers(0, 256))
            elif mode == 1: a[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 2: del a[pos:pos + int(rng.integers(1, 9))]
            else: a[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8))
        p = os.path.join(d, "f." + ext)
        with open(p, "wb") as f: f.write(a)
        try:
            rrt.decode_image_file(p); ok += 1
        except rrt.RrtError:
            err += 1
print(f"{n_iter} mutated files: {ok} decoded, {err} refused, no crash")
