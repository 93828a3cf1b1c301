#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/ keeps for one library build, in one call - the full GPU test
# log, the bench lines of every workload with the kernel summary and timeline of the default one, the 2-rank
# rehearsal line, and the counter passes.  Results land under gpurun_out/final/ (and gpurun_out/refresh, pmc, pmc_sq).
#   STAGE=1: tests, bench lines of c3 / c2 / c4 with kernel summary and timeline; STAGE=2: the rest (default: both; one
#   gpurun call is limited to 20 minutes)
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT"
cd "$ROOT"
if [ "${STAGE:-1}" = 1 ] || [ -z "${STAGE:-}" ]; then
timeout -k 10 900 python3 -u -m pytest tests -m gpu -v --timeout 180 --timeout-method thread > "$OUT/gpu_tests_full.txt" 2>&1 || echo "GPU TESTS FAILED"
tail -1 "$OUT/gpu_tests_full.txt"
bash tools/refresh_profiles.sh > "$OUT/refresh.log" 2>&1
fi
if [ "${STAGE:-}" = 1 ]; then echo done stage 1; exit 0; fi
for w in clean clean_cfg clean_h1; do
    timeout -k 10 400 python3 bench.py --workload $w > "$OUT/bench_$w.log" 2>&1
    grep -o '{"metric.*' "$OUT/bench_$w.log" > "$OUT/bench_$w.json"
done
DANRA_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --timesteps 200 --no-cpu-baseline > "$OUT/rehearsal.log" 2>&1
grep -o '{"metric.*' "$OUT/rehearsal.log" > "$OUT/bench_rehearsal_2rank.json"
DANRA_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload c4 --steps 1 --warmup 1 --timesteps 100 --no-cpu-baseline --no-strict-fp32 > "$OUT/rehearsal_c4.log" 2>&1
grep -o '{"metric.*' "$OUT/rehearsal_c4.log" > "$OUT/bench_rehearsal_2rank_c4.json"
bash tools/pmc_traffic.sh > "$OUT/pmc_traffic.log" 2>&1
bash tools/pmc_sq.sh > "$OUT/pmc_sq.log" 2>&1
echo done
