cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 900 python -m pytest tests/test_modules_gpu.py tests/test_round2_gpu.py -q -m gpu 2>&1 | tail -4
for OV in 0 1; do for B in 16 64 256; do
  KALLE_OVERLAP_ADAM=$OV timeout -k 10 300 python bench.py --no-cpu-baseline --batch $B --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$OV', d['config']['per_gpu_batch'], round(d['ms_per_step'],2), round(d['value']), round(d['algorithmic_tflops_per_gpu']), 'loss', d['config']['loss'], d['comm'])"
done; done
