#!/bin/bash
# Round 6 evidence on a GPU box (one gpurun call each part; run from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'PART=1 bash tools/profile_round6.sh'     bench line + kernel traces (f64 headline, f32) + FETCH / WRITE
#   gpurun --timeout 1200 -- 'PART=2 bash tools/profile_round6.sh'     the other bench shapes and probes (PART=3: configs kernel trace, PART=4: fuzz, soak, presets, B = 1)
# then, back in the build container (see tools/prof_summary.py):
#   python tools/prof_summary.py --round r06 --tag f64 --kt gpurun_out/prof_kt_f64 --fetch gpurun_out/prof_fetch_f64 \
#       --write gpurun_out/prof_write_f64 --key k_actor_streamed_3wrobot_B65536_K256_N10_f64
#   python tools/prof_summary.py --round r06 --tag f32 --kt gpurun_out/prof_kt_f32 --fetch gpurun_out/prof_fetch_f32 \
#       --write gpurun_out/prof_write_f32 --key k_actor_streamed_3wrobot_B65536_K256_N10_f32
# The headline of bench.py is the float64 tick since round 6 (the reference's arithmetic width); the program itself follows `--`
# (no env / bash -c hop under rocprofv3), counters in passes of their own (MI355X_MICROARCH.md, HBM / rocprofv3 section).
set -u
PART=${PART:-1}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
B="--no-cpu-baseline --no-secondary"
if [ "$PART" = 1 ]; then
python bench.py > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err
for dt in f64 f32; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt_$dt -o kt -- \
  python3 bench.py --dtype $dt $B > gpurun_out/prof_kt_$dt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch_$dt -o f -- \
  python3 bench.py --dtype $dt --steps 20 --warmup 3 $B --no-parity > gpurun_out/prof_fetch_$dt.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write_$dt -o w -- \
  python3 bench.py --dtype $dt --steps 20 --warmup 3 $B --no-parity > gpurun_out/prof_write_$dt.log 2>&1
done
# the driver's form, three consecutive runs
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_driver_form_1.json 2>> gpurun_out/bench_r06.err
python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_driver_form_2.json 2>> gpurun_out/bench_r06.err
python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_driver_form_3.json 2>> gpurun_out/bench_r06.err
fi
if [ "$PART" = 2 ]; then
O=gpurun_out/ev_r06
mkdir -p $O
python bench.py --config C3 $B > $O/r06_bench_c3_f64.json 2> $O/err.log
python bench.py --config C3 --dtype f32 $B > $O/r06_bench_c3_f32.json 2>> $O/err.log
python bench.py --config C3 --dtype f32 --tick-parts 0 $B > $O/r06_bench_c3_f32_unsplit_default_on_callers_stream.json 2>> $O/err.log
python bench.py --config C4 --steps 50 --warmup 10 $B > $O/r06_bench_c4_one_rank_f64.json 2>> $O/err.log
python bench.py --config C5 $B > $O/r06_bench_c5.json 2>> $O/err.log
python bench.py --regime generated $B > $O/r06_bench_c2_generated.json 2>> $O/err.log
python bench.py --nactor 20 $B > $O/r06_bench_c2_nactor20_f64.json 2>> $O/err.log
python bench.py --force-dist --steps 100 --warmup 10 $B > $O/r06_force_dist_rccl_world1.json 2>> $O/err.log
python bench.py --gpus 2 --dist-backend gloo --single-device --steps 100 --warmup 10 $B > $O/r06_two_ranks_gloo_one_gpu.json 2>> $O/err.log
python tools/opt_probe.py 2>/dev/null | grep -v amdgpu > $O/r06_opt_probe.txt
python tools/critic_stream_probe.py f32 2>/dev/null | grep -v amdgpu > $O/r06_critic_stream_probe_f32.txt
python tools/critic_stream_probe.py f64 2>/dev/null | grep -v amdgpu > $O/r06_critic_stream_probe_f64.txt
python tools/generic_stream_probe.py f32 2>/dev/null | grep -v amdgpu > $O/r06_generic_stream_probe_f32.txt
python tools/generic_stream_probe.py f64 2>/dev/null | grep -v amdgpu > $O/r06_generic_stream_probe_f64.txt
ls $O
fi
if [ "$PART" = 3 ]; then
# kernel trace of the configs that are parity cases rather than the bench line (configs[2] generated / streamed x MPC / RQL / SQL,
# optimiser tick, operator boundary, nominal ticks, disturbed tick, env step at 2^24 envs, configs[4] shard): k_critic_fit et al.
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt_configs_r06 -o kt -- \
  python3 tools/bench_configs.py > gpurun_out/bench_configs_r06_profiled.json 2> gpurun_out/prof_kt_configs_r06.log
python tools/bench_configs.py > gpurun_out/bench_configs_r06.json 2> gpurun_out/bench_configs_r06.err
fi
if [ "$PART" = 4 ]; then
# the randomised differential run (oracle as the checker) and the soak
python tools/fuzz_parity.py 1500 31337 2>/dev/null | grep -v amdgpu | tail -4 > gpurun_out/r06_fuzz_tail.txt
python tools/soak_r06.py 2>/dev/null | grep -v amdgpu > gpurun_out/r06_soak.txt
python tools/preset_timing.py 2>/dev/null | grep -v amdgpu > gpurun_out/r06_preset_timing.txt
python tools/b1_profile.py 2>/dev/null | grep -v amdgpu | head -4 > gpurun_out/r06_b1_profile_head.txt
cat gpurun_out/r06_fuzz_tail.txt gpurun_out/r06_b1_profile_head.txt
fi
