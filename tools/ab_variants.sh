# Developer tool (run on the GPU box): kernel time of the three single-GPU workloads for each given library variant suffix ("" = the product).
#   gpurun -- bash tools/ab_variants.sh "" _f_o2 _f_exh ...      (a suffix may also be a path to a library file)
for v in "$@"; do
  lib=rust-ray-tracer_amd/librrt_hip$v.so; [ -f "$v" ] && lib=$v
  [ -f $lib ] || { echo "missing $v"; continue; }
  line="prod$v"
  for a in "" "--scene soup100000" "--scene soup1000000 --width 3840 --height 2160"; do
    r=$(RRT_LIB=$PWD/$lib timeout -k 5 160 python bench.py $a --steps 20 --no-cpu-baseline --no-host-fb --no-first-frame 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['kernel_ms'], d['config']['filter_variant'][0], d['frame_checksum'] % 100000)" 2>/dev/null || echo "FAIL")
    line="$line | $r"
  done
  echo "$line"
done
