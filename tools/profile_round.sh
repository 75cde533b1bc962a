#!/bin/bash
# Reproduce the evidence kept under profiles/ for one round, on a GPU box (run through gpurun from the repo root):
#
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01_final'
#   python tools/prof_summary.py --round r01_final --kt gpurun_out/prof_kt --fetch gpurun_out/prof_fetch \
#       --write gpurun_out/prof_write --key k_actor_streamed_B65536_K256_N10_f32      # back in the build container
#
# Passes (MI355X_MICROARCH.md, HBM / rocprofv3 section): the kernel trace of the SAME command the bench line comes from,
# then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (never combined with other trace domains).  The program itself
# follows `--` (no env / bash -c hop: the profiler's preloaded library has already initialised the GPU).
set -u
ROUND=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python bench.py > "gpurun_out/bench_${ROUND}.json" 2> "gpurun_out/bench_${ROUND}.err"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- \
  python bench.py --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o f -- \
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o w -- \
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/prof_write.log 2>&1
python tools/bench_configs.py --steps 30 > "gpurun_out/bench_configs_${ROUND}.json" 2> gpurun_out/bench_configs.err
ls gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
head -c 600 "gpurun_out/bench_${ROUND}.json"
