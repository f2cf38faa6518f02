#!/bin/bash
# un-profiled bench lines of every single-GPU configuration (copied to profiles/ as rNN_bench_<config>.json)
python bench.py > gpurun_out/bench_cfg4_100mp.json 2> gpurun_out/bench_cfg4.err
python bench.py --config cfg3_45mp --no-cpu-baseline > gpurun_out/bench_cfg3_45mp.json 2>/dev/null
python bench.py --config cfg2_24mp --no-cpu-baseline > gpurun_out/bench_cfg2_24mp.json 2>/dev/null
python bench.py --config cfg5_batch --no-cpu-baseline > gpurun_out/bench_cfg5_batch.json 2>/dev/null
python bench.py --no-graph --no-cpu-baseline --no-pcie > gpurun_out/bench_cfg4_nograph.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --same-device --no-cpu-baseline --no-alone 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_2ranks_one_gpu.json
python tools/shard_model.py 2>&1 | grep -v amdgpu.ids > gpurun_out/shard_model.txt
R2F_SHARD_SPLIT=0 python tools/shard_model.py 2>&1 | grep -v amdgpu.ids > gpurun_out/shard_model_nosplit.txt
