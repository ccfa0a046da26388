#!/bin/bash
# round 2, profile visit: the whole GPU suite, the driver's bench command plain and under rocprofv3, PMC passes, every configuration
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v8
rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-400 $O/bench_driver.json
RSRL_NO_COALESCE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-shared-leg --no-streaming-leg > $O/bench_k20_nocoalesce.json 2>&1
timeout 300 python scripts/bench_configs.py > $O/bench_configs.jsonl 2>&1; cut -c1-200 $O/bench_configs.jsonl
timeout 300 python scripts/scale_n.py > $O/scale_n.jsonl 2>&1; cat $O/scale_n.jsonl
for e in none rccl peer; do python scripts/prof_shared.py fourier $e; done > $O/shared_walls.txt 2>&1
python scripts/prof_shared.py tile none >> $O/shared_walls.txt 2>&1; grep us/step $O/shared_walls.txt
cd /tmp
FU="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps 2560 --warmup 256 --repeats 4 --steps-per-launch 256"   # the PMC passes are normalised per 256-step launch
K20="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps 20 --warmup 5 --repeats 200"
K1="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps-per-launch 1 --steps 400 --warmup 100 --repeats 2"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/fu_$i -o p -- $FU > $O/fu_$i.log 2>&1
  if [ $i -ge 2 ]; then timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/k1_$i -o p -- $K1 > $O/k1_$i.log 2>&1; fi
  if [ $i -ge 3 ]; then RSRL_NO_COALESCE=1 timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/k20_$i -o p -- $K20 > $O/k20_$i.log 2>&1; fi
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_driver -o b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_prof.json 2> $O/stats_driver.log
for e in none rccl peer; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fourier_$e -o s -- python $GRAFT_REPO_ROOT/scripts/prof_shared.py fourier $e > $O/prof_fourier_$e.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tile_none -o s -- python $GRAFT_REPO_ROOT/scripts/prof_shared.py tile none > $O/prof_tile.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -o s -- python $GRAFT_REPO_ROOT/scripts/bench_configs.py "C5/2" "C5'" "L1" "C3'" > $O/prof_configs.log 2>&1
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*_agent_info.csv" -delete
du -sh $O
