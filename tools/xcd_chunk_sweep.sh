# Developer tool (run on the GPU box): frame time of the single-GPU workloads for block-order chunk sizes (RRT_XCD_CHUNK, blocks per chunk; 0 = plain order).
#   gpurun -- bash tools/xcd_chunk_sweep.sh 0 64 256 1024
for c in "$@"; do
  line="chunk $c"
  for a in "" "--walk lane" "--scene assets/model3.obj" "--scene soup100000" "--scene soup1000000 --width 3840 --height 2160"; do
    r=$(RRT_XCD_CHUNK=$c timeout -k 5 160 python bench.py $a --steps 20 --no-cpu-baseline --no-host-fb --no-first-frame 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['kernel_ms'], d['config']['filter_variant'][0], d['frame_checksum'] % 100000)" 2>/dev/null || echo "FAIL")
    line="$line | $r"
  done
  echo "$line"
done
