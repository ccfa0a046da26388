#!/bin/bash
# round-3 evidence: the driver's bench invocation plain and under rocprofv3 (kernel stats), then per-kernel stats + PMC passes of
# the shared-W persistent kernel, the tile-coding step and the streaming kernel.  Outputs under gpurun_out/kprof/ and gpurun_out/r03/.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r03/bench_driver.json 2> $R/gpurun_out/r03/bench_driver.err
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS"
# the same command under the profiler (the profiler slows the host: the library coalesces the 20-step calls into shorter launches than in
# the plain run -- compare per batch-step), and with 1024-step calls, where a call is exactly one launch
bash $R/scripts/gpu_kprof.sh bench_driver -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-legs --no-shared-leg --no-streaming-leg > $R/gpurun_out/r03/bench_driver_profiled.json 2>/dev/null
bash $R/scripts/gpu_kprof.sh bench_1024 -- python $R/bench.py --gpus 1 --steps 1024 --warmup 5 --no-cpu-baseline --no-config-legs --no-shared-leg --no-streaming-leg --no-nocoalesce-leg > $R/gpurun_out/r03/bench_1024_profiled.json 2>/dev/null
python $R/bench.py --gpus 1 --steps 1024 --warmup 5 --no-cpu-baseline --no-config-legs --no-shared-leg --no-streaming-leg --no-nocoalesce-leg > $R/gpurun_out/r03/bench_1024.json 2>/dev/null
bash $R/scripts/gpu_kprof.sh legs -- python $R/bench.py --gpus 1 --steps 1024 --warmup 5 --no-cpu-baseline --regions 1 --region-seconds 0.2 --no-nocoalesce-leg > /dev/null 2>&1
bash $R/scripts/gpu_kprof.sh shared_persist --pmc "$SQ1" --pmc "$SQ2" --pmc "FETCH_SIZE" --pmc "WRITE_SIZE" -- python $R/scripts/prof_shared.py fourier none > /dev/null 2>&1
bash $R/scripts/gpu_kprof.sh shared_persist_peer -- python $R/scripts/prof_shared.py fourier peer > /dev/null 2>&1
RSRL_NO_PERSIST=1 bash $R/scripts/gpu_kprof.sh shared_perstep -- python $R/scripts/prof_shared.py fourier none > /dev/null 2>&1
bash $R/scripts/gpu_kprof.sh tile --pmc "$SQ1" -- python $R/scripts/prof_shared.py tile none > /dev/null 2>&1
python $R/scripts/bench_configs.py > $R/gpurun_out/r03/bench_configs.jsonl 2>/dev/null
for t in bench_driver bench_1024 legs shared_persist shared_persist_peer shared_perstep tile; do echo "== $t"; head -6 $R/gpurun_out/kprof/$t/stats.txt | cut -c1-200; done
