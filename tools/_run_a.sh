timeout -s KILL 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_step.py tests/test_gpu_kernels.py tests/test_examples.py -m gpu -q --tb=short -W ignore 2>&1 | grep -E 'passed|failed|Error|assert|FAILED' | tail -12
for w in cnn_simple_b256 cnn_reference_b256; do python bench.py --no-cpu-baseline --no-roofline --workload $w --steps 400 --warmup 40 | tail -1 | cut -c1-200; done
