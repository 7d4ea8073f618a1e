timeout 800 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_n1.json; cat gpurun_out/bench_n1.json
