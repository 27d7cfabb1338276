import importlib,os,sys,time
sys.path.insert(0,os.getcwd())
rrt=importlib.import_module("rust-ray-tracer_amd"); syn=importlib.import_module("rust-ray-tracer_amd.synthetic")
A=os.path.join(os.getcwd(),"assets")
for name,path,w,h in (("teapot",os.path.join(A,"model2.obj"),1920,1080),("soup100k",syn.ensure_soup(A,100000,syn.SEED_100K),1920,1080),("soup1m",syn.ensure_soup(A,1000000,syn.SEED_1M),3840,2160)):
    for r in range(4):
        t0=time.perf_counter(); sd=rrt.parse_obj_file(path); t1=time.perf_counter(); rt=rrt.RayTracer(sd,rrt.default_lights()); t2=time.perf_counter(); rt.render(w,h); t3=time.perf_counter()
        print(f"{name} rep {r}: load {1e3*(t1-t0):.2f} create {1e3*(t2-t1):.2f} render {1e3*(t3-t2):.2f} total {1e3*(t3-t0):.2f} ms", rt.setup_times() if r==3 else "", flush=True)
        del rt, sd
