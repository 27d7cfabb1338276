set -e
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
for a in "" "--scene soup100000" "--scene soup1000000 --width 3840 --height 2160"; do
python bench.py $a --steps 20 --no-cpu-baseline --no-host-fb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['config']['workload'][:30], d['kernel_ms'], d['config'].get('filter_variant'), d['frame_checksum'])"
done
