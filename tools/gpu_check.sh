# Developer tool: GPU suite + the three single-GPU bench workloads, one line each (run on the GPU box: gpurun -- bash tools/gpu_check.sh [variant ...])
set -e
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
for v in "${@:-auto}"; do
for a in "" "--scene soup100000" "--scene soup1000000 --width 3840 --height 2160"; do
python bench.py $a --walk $v --steps 20 --no-cpu-baseline --no-host-fb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['config']['workload'][:30], d['kernel_ms'], d['config'].get('filter_variant'), d['frame_checksum'])"
done; done
