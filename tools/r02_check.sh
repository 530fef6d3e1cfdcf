#!/bin/bash
# round-2 GPU check: parity tests, the bench line (N=1, with secondary + CPU legs), the
# self-launched 2-rank bench (test aid: both ranks on the one GPU of the box, gloo)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02_pytest.log
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
python bench.py --gpus 2 --oversubscribe --steps 10 --no-cpu > gpurun_out/r02_bench_2rank.json 2> gpurun_out/r02_bench_2rank.err
tail -5 gpurun_out/r02_pytest.log
tail -3 gpurun_out/r02_bench.err
head -c 3000 gpurun_out/r02_bench.json
echo
tail -5 gpurun_out/r02_bench_2rank.err
head -c 1500 gpurun_out/r02_bench_2rank.json
