#!/bin/bash
# round 3, visit 1: baseline -- GPU suite duration + the driver's bench invocation at HEAD of round 2
set -u
mkdir -p gpurun_out/r3v1
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/r3v1/pytest.log 2>&1
tail -30 gpurun_out/r3v1/pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3v1/bench.json 2> gpurun_out/r3v1/bench.err
tail -3 gpurun_out/r3v1/bench.err; cut -c1-600 gpurun_out/r3v1/bench.json
