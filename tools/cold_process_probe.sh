# Developer tool (GPU box): the FIRST load + create + render of a fresh process (what a one-shot host pays), three processes in a row
for k in 1 2 3; do
env ${TRACE:+RRT_SETUP_TRACE=1} python - <<'PY' 2>&1 | grep -v amdgpu
import importlib, os, sys, time
t_imp = time.perf_counter()
sys.path.insert(0, os.getcwd())
rrt = importlib.import_module("rust-ray-tracer_amd")
rrt.lib()
t0 = time.perf_counter(); sd = rrt.parse_obj_file("assets/model2.obj"); t1 = time.perf_counter(); rt = rrt.RayTracer(sd, rrt.default_lights()); t2 = time.perf_counter(); rt.render(1920, 1080); t3 = time.perf_counter()
print(f"cold process: import+dlopen {1e3*(t0-t_imp):.1f}  load {1e3*(t1-t0):.2f}  create {1e3*(t2-t1):.2f}  render {1e3*(t3-t2):.2f}  total {1e3*(t3-t0):.2f} ms   {rt.setup_times()['hip_init_ms']:.1f} ms hip init on the caller")
PY
done
