#!/bin/bash
# NOTE (round 6): bench.py's default element type is now f64 (the reference's width); this script was written for the f32 default of
# rounds 1-5 - pass --dtype f32 where it says nothing, or use tools/profile_round6.sh, which produced profiles/r06_*.
# Reproduce the evidence kept under profiles/ for one round, on a GPU box (run through gpurun from the repo root):
#
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03'
#   python tools/prof_summary.py --round r03 --kt gpurun_out/prof_kt --fetch gpurun_out/prof_fetch \
#       --write gpurun_out/prof_write --key k_actor_streamed_3wrobot_B65536_K256_N10_f32 \
#       --valu gpurun_out/prof_valu --valu-units gpurun_out/valu_units.json          # back in the build container
#   python tools/prof_summary.py --round r03_ticks --valu gpurun_out/prof_valu_ticks --valu-units gpurun_out/valu_units_ticks.json
#   python tools/prof_summary.py --round r03_pool --valu gpurun_out/prof_valu_pool --valu-units gpurun_out/valu_units_pool.json
#   python tools/prof_summary.py --round r03_c3rql --valu gpurun_out/prof_valu_c3rql --valu-units gpurun_out/valu_units_c3rql.json
#   python tools/prof_summary.py --round r03 --tag configs --kt gpurun_out/prof_kt_configs
#   for t in c2_f64 c2_n20 c3_f32 c3_f64: python tools/prof_summary.py --round r03 --tag $t --fetch gpurun_out/prof_fetch_$t \
#       --write gpurun_out/prof_write_$t --key <the key bench.py looks up for that shape>    (tools/profile_keys.sh prints them)
#
# Passes (MI355X_MICROARCH.md, HBM / rocprofv3 section): the kernel trace of the SAME command the bench line comes from,
# then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (never combined with other trace domains), then the SQ counters
# of the VALU-bound kernels (tools/valu_probe.py), then the kernel trace of tools/bench_configs.py (configs[2], [4]).
# The program itself follows `--` (no env / bash -c hop: the profiler's preloaded library has already initialised the GPU).
# A gpurun call lasts at most 1200 s: PART=1 (bench line, kernel trace, FETCH / WRITE of the bench shape, the SQ passes) and
# PART=2 (FETCH / WRITE of the other shapes, tools/bench_configs.py) split the work over two calls; default: everything.
set -u
ROUND=${1:-r02}
PART=${PART:-all}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
if [ "$PART" != 2 ]; then
python bench.py > "gpurun_out/bench_${ROUND}.json" 2> "gpurun_out/bench_${ROUND}.err"
# (--no-secondary: the secondary regimes include the SAME kernel at half the batch - two handles on two streams - which
# would blend two launch sizes into one average)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- \
  python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o f -- \
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o w -- \
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity > gpurun_out/prof_write.log 2>&1
fi
# FETCH_SIZE / WRITE_SIZE of the other streamed shapes bench.py can be asked for (roofline.traffic of --dtype f64, --nactor 20,
# --config C3 in both element types; the C3 passes also hold k_critic_fit, the big-batch sim pass k_sim_v)
pmc_pair() {  # tag, bench arguments
  local tag=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "gpurun_out/prof_fetch_${tag}" -o f -- \
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity "$@" > "gpurun_out/prof_fetch_${tag}.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "gpurun_out/prof_write_${tag}" -o w -- \
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity "$@" > "gpurun_out/prof_write_${tag}.log" 2>&1
}
if [ "$PART" != 1 ]; then
pmc_pair c2_f64 --dtype f64
pmc_pair c2_n20 --nactor 20
pmc_pair c3_f32 --config C3
pmc_pair c3_f64 --config C3 --dtype f64
fi
if [ "$PART" != 2 ]; then
SQ="SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/prof_valu -o v -- \
  python3 tools/valu_probe.py main > gpurun_out/valu_units.json 2> gpurun_out/prof_valu.log
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/prof_valu_ticks -o v -- \
  python3 tools/valu_probe.py ticks > gpurun_out/valu_units_ticks.json 2> gpurun_out/prof_valu_ticks.log
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/prof_valu_pool -o v -- \
  python3 tools/valu_probe.py pool > gpurun_out/valu_units_pool.json 2> gpurun_out/prof_valu_pool.log
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/prof_valu_c3rql -o v -- \
  python3 tools/valu_probe.py c3rql > gpurun_out/valu_units_c3rql.json 2> gpurun_out/prof_valu_c3rql.log
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/prof_valu_search -o v -- \
  python3 tools/valu_probe.py search > gpurun_out/valu_units_search.json 2> gpurun_out/prof_valu_search.log
fi
if [ "$PART" != 1 ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt_configs -o kt -- \
  python3 tools/bench_configs.py > "gpurun_out/bench_configs_${ROUND}_profiled.json" 2> gpurun_out/prof_kt_configs.log
python tools/bench_configs.py > "gpurun_out/bench_configs_${ROUND}.json" 2> gpurun_out/bench_configs.err
fi
ls gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_valu gpurun_out/prof_kt_configs
head -c 600 "gpurun_out/bench_${ROUND}.json"
