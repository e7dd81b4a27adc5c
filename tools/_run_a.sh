timeout -s KILL 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_step.py -m gpu -q --tb=short -W ignore 2>&1 | grep -E 'passed|failed|Error|assert|FAILED' | tail -12
for b in 4096 16384 60000; do python bench.py --no-cpu-baseline --no-roofline --no-sweep --batch $b --steps 300 --warmup 20 | tail -1 | cut -c1-160; done
