# Developer tool (GPU box): kernel time of 2^20 random rays (rrt_intersect_rays) per traversal variant, for each library variant suffix
for v in "$@"; do
  for sc in soup100000 model; do
    a=$([ $sc = model ] && echo "" || echo $sc)
    r=$(RRT_LIB=rust-ray-tracer_amd/librrt_hip$v.so python tools/random_rays_probe.py $a 2>/dev/null | awk '{printf "%s %s | ", $1, $6}')
    echo "prod$v $sc: $r"
  done
done
