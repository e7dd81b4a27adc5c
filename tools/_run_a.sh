timeout -s KILL 300 python -m pytest tests/test_gpu_mlp_tail.py tests/test_gpu_fused.py -m gpu -q --tb=short -W ignore 2>&1 | grep -E 'passed|failed' | tail -3
for g in 0 1; do TAPER_TAIL_GENERAL=$g python bench.py --no-sweep --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [(k['kernel'][:18],k['us_per_launch']) for k in d['roofline']['kernels']], d['roofline']['step_us_two_launch_chain'])
"; done
