set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in cfg4_100mp cfg3_45mp cfg2_24mp; do bash tools/profile_round.sh r02 $c > gpurun_out/profile_r02_$c.log 2>&1; done
python bench.py > gpurun_out/bench_cfg4_100mp.json 2> gpurun_out/bench_cfg4.err
python bench.py --config cfg3_45mp --no-cpu-baseline > gpurun_out/bench_cfg3_45mp.json 2>/dev/null
python bench.py --config cfg2_24mp --no-cpu-baseline > gpurun_out/bench_cfg2_24mp.json 2>/dev/null
python bench.py --config cfg5_batch --no-cpu-baseline > gpurun_out/bench_cfg5_batch.json 2>/dev/null
python tools/parity_budget.py > gpurun_out/parity_budget.txt 2>&1
python bench.py > gpurun_out/bench_cfg4_100mp_b.json 2>/dev/null
