"""Developer tool: decode every JPEG under assets/ through rrt_decode_image_file, print a digest and the time (RRT_SETUP_TRACE=1 for the phases, RRT_JPEG_SERIAL=1 for the one-thread entropy decode)."""
import importlib,time,sys,os,glob,hashlib
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rrt=importlib.import_module("rust-ray-tracer_amd")
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'assets', '*.jpg'))):
    t=time.perf_counter(); a=rrt.decode_image_file(f); dt=(time.perf_counter()-t)*1e3
    print(f.split('/')[-1], a.shape, hashlib.sha1(a.tobytes()).hexdigest()[:12], f"{dt:.1f} ms")
