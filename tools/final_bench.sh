#!/bin/bash
# un-profiled bench lines of every single-GPU configuration and variant (copied to profiles/ as rNN_bench_<name>.json), the one-GPU
# shard model with and without the exposure-range record, and the interleaved A/B of the device-side scratch-element choice
mkdir -p gpurun_out/final
O=gpurun_out/final
python bench.py > $O/bench_cfg4_100mp.json 2> $O/bench_cfg4.err
python bench.py --config cfg3_45mp --no-cpu-baseline > $O/bench_cfg3_45mp.json 2>/dev/null
python bench.py --config cfg2_24mp --no-cpu-baseline > $O/bench_cfg2_24mp.json 2>/dev/null
python bench.py --config cfg5_batch --no-cpu-baseline > $O/bench_cfg5_batch.json 2>/dev/null
python bench.py --no-graph --no-cpu-baseline --no-pcie > $O/bench_cfg4_nograph.json 2>/dev/null
python bench.py --output u8 --no-cpu-baseline --no-pcie > $O/bench_cfg4_u8_output.json 2>/dev/null
python bench.py --frame smooth --no-cpu-baseline --no-pcie > $O/bench_cfg4_smooth.json 2>/dev/null
python bench.py --clamp 0.004,48 --no-cpu-baseline --no-pcie > $O/bench_cfg4_clamped.json 2>/dev/null
python bench.py --opt stencil_fft_scratch96_auto=0 --no-cpu-baseline --no-pcie > $O/bench_cfg4_complex128.json 2>/dev/null
python bench.py --checksum --no-cpu-baseline --no-pcie --no-alone > $O/bench_cfg4_checksum.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --same-device --no-cpu-baseline --no-alone --checksum 2>/dev/null | tail -1 > $O/bench_cfg4_2ranks_one_gpu.json
python bench.py --gpus 8 --backend gloo --same-device --no-cpu-baseline --no-alone --checksum 2>/dev/null | tail -1 > $O/bench_cfg4_8ranks_one_gpu.json
python tools/shard_model.py 2>&1 | grep -v amdgpu.ids > $O/shard_model.txt
R2F_SHARD_DYN=0 python tools/shard_model.py 2>&1 | grep -v amdgpu.ids > $O/shard_model_no_range_record.txt
python tools/ab_render.py --set stencil_fft_scratch96_auto=1 --set stencil_fft_scratch96_auto=0 2>&1 | grep -v amdgpu.ids > $O/ab_scratch_choice.txt
