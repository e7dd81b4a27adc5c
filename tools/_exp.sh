cd $GRAFT_REPO_ROOT
python bench.py --workloads cnn_reference_b256 --no-sweep --no-cpu-baseline --no-roofline --steps 100 --warmup 10 > /dev/null 2>&1; python -c "
import json; d=json.load(open('gpurun_out/bench_details.json'))
for w in d['workloads']:
    for k in w.get('kernels',[]): print('   ', k.get('layer',''), k['us_per_launch'], k['frac'], k.get('in_step'))
"
TAPER_CONV_LAYER_CHAIN=0 python bench.py --workloads cnn_reference_b256 --no-sweep --no-cpu-baseline --no-roofline --steps 100 --warmup 10 > /dev/null 2>&1; python -c "
import json; d=json.load(open('gpurun_out/bench_details.json'))
for w in d['workloads']:
    for k in w.get('kernels',[]): print('  old', k.get('layer',''), k['us_per_launch'], k['frac'], k.get('in_step'))
"
