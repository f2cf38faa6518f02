#!/bin/bash
# tools/first_multi_gpu.sh [tag] -- ONE command for the first box with more than one MI355X.
# No round of this build had such a box: RCCL has not carried a byte, DESIGN.md section 5's scaling table is a one-GPU model.
# Run from the repository root on an N-GPU node (N = 2, 4 or 8; larger counts than the node has are skipped):
#   1. the RCCL tests (tests/test_gpu_multi.py: send / recv pairs over xGMI, checksum equality, one processor per device);
#   2. bench.py --gpus {1,2,4,8} --checksum      -> the same frame on every N: the checksums must agree to the FFT's rounding
#   3. bench.py --gpus {1,2,4,8}                 -> the four JSON lines of the 1 -> 8 curve (rank 0 prints one line each)
# and collect, per N: rccl_ranks, the schedule the ranks measured (config.shard_schedule.measured_ms_max_over_ranks -- the
# exchange included: the constant that replaces DESIGN.md section 5's modelled column), ms_per_step and the checksum into
# profiles/<tag>_multi_gpu_summary.txt, next to the raw lines profiles/<tag>_bench_cfg4_gpus<N>[_checksum].json.
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r07}
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
mkdir -p profiles gpurun_out
echo "first_multi_gpu: $NGPU GPU(s) visible"
python -m pytest tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -5 | tee profiles/${TAG}_multi_gpu_tests.txt
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && continue
  python bench.py --gpus $N --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-alone --checksum 2> gpurun_out/${TAG}_gpus${N}_checksum.err | tail -1 > profiles/${TAG}_bench_cfg4_gpus${N}_checksum.json
  python bench.py --gpus $N --steps 20 --warmup 3 --no-cpu-baseline --no-pcie --no-alone 2> gpurun_out/${TAG}_gpus${N}.err | tail -1 > profiles/${TAG}_bench_cfg4_gpus${N}.json
done
python - "$TAG" <<'PY' | tee profiles/${TAG}_multi_gpu_summary.txt
import glob, json, sys
tag = sys.argv[1]
print(f"# tools/first_multi_gpu.sh {tag}: 100 MP frame row-sharded over N MI355X (RCCL halo exchange over xGMI), strong scaling")
print("#  N  ms_per_step      MP/s  vs N=1  efficiency  rccl_ranks  schedule (exchanges, interior halation first)  measured ms per candidate (max over ranks)  checksum")
base = None
for n in (1, 2, 4, 8):
    try:
        d = json.load(open(f"profiles/{tag}_bench_cfg4_gpus{n}.json"))
        c = json.load(open(f"profiles/{tag}_bench_cfg4_gpus{n}_checksum.json"))
    except (OSError, ValueError):
        continue
    base = base or d["ms_per_step"]
    s = d["config"].get("shard_schedule", {})
    print(f"  {n:2d}  {d['ms_per_step']:10.3f}  {d['value']:9.0f}  {base / d['ms_per_step']:6.2f}  {base / d['ms_per_step'] / n:10.2f}  "
          f"{d.get('rccl_ranks', '-'):>10}  {(s.get('exchanges'), s.get('interior_halation_ahead_of_the_exchange'))!s:>20}  "
          f"{s.get('measured_ms_max_over_ranks')}  {c.get('checksum')}")
PY
