#!/bin/bash
# NOTE (round 6): bench.py's default element type is now f64 (the reference's width); this script was written for the f32 default of
# rounds 1-5 - pass --dtype f32 where it says nothing, or use tools/profile_round6.sh, which produced profiles/r06_*.
# The measurements DESIGN.md section 5 quotes beyond tools/profile_round.sh, written under gpurun_out/ev_<round>/ on a GPU box:
#   gpurun --timeout 1200 -- 'bash tools/evidence_round.sh r03'   then copy gpurun_out/ev_r03/* to profiles/ (tracked)
set -u
R=${1:-r03}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ev_${R}
mkdir -p "$O"
B="--no-cpu-baseline --no-secondary"
PART=${PART:-all}   # a gpurun call lasts at most 1200 s: PART=1 bench lines, PART=2 sweeps / probes / test output
if [ "$PART" != 2 ]; then
# the driver's form, three consecutive runs (default flags otherwise: CPU baseline and secondary regimes included in the first)
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/${R}_driver_form_1.json" 2> "$O/err.log"
python bench.py --gpus 1 --steps 20 --warmup 5 $B > "$O/${R}_driver_form_2.json" 2>> "$O/err.log"
python bench.py --gpus 1 --steps 20 --warmup 5 $B > "$O/${R}_driver_form_3.json" 2>> "$O/err.log"
# RCCL at world size 1; two ranks over gloo on the one GPU (functional: the N > 1 code path)
python bench.py --force-dist --steps 100 --warmup 10 $B > "$O/${R}_force_dist_rccl_world1.json" 2>> "$O/err.log"
python bench.py --gpus 2 --dist-backend gloo --single-device --steps 100 --warmup 10 $B > "$O/${R}_two_ranks_gloo_one_gpu.json" 2>> "$O/err.log"
# the other configs / shapes as bench lines
python bench.py --parts 2 $B > "$O/${R}_bench_c2_two_handles.json" 2>> "$O/err.log"
python bench.py --config C3 $B > "$O/${R}_bench_c3_two_handles.json" 2>> "$O/err.log"
python bench.py --config C3 --parts 1 $B > "$O/${R}_bench_c3_one_handle.json" 2>> "$O/err.log"
python bench.py --config C3 --parts 1 --tick-parts 1 $B > "$O/${R}_bench_c3_one_handle_unsplit.json" 2>> "$O/err.log"
python bench.py --config C3 --dtype f64 $B > "$O/${R}_bench_c3_f64.json" 2>> "$O/err.log"
python bench.py --config C3 --regime generated $B > "$O/${R}_bench_c3_generated.json" 2>> "$O/err.log"
python bench.py --config C4 --steps 50 --warmup 10 $B > "$O/${R}_bench_c4_one_rank.json" 2>> "$O/err.log"
python bench.py --config C5 $B > "$O/${R}_bench_c5.json" 2>> "$O/err.log"
python bench.py --dtype f64 $B > "$O/${R}_bench_c2_f64.json" 2>> "$O/err.log"
python bench.py --nactor 20 $B > "$O/${R}_bench_c2_nactor20.json" 2>> "$O/err.log"
python bench.py --regime generated $B > "$O/${R}_bench_c2_generated.json" 2>> "$O/err.log"
fi
if [ "$PART" != 1 ]; then
for k in 8 16 32 36 48 130; do python bench.py --candidates $k --steps 300 --warmup 30 $B > "$O/${R}_bench_c2_K${k}.json" 2>> "$O/err.log"; done
# probes
python tools/critic_stream_probe.py f32 2>/dev/null | grep -v amdgpu > "$O/${R}_critic_stream_probe_f32.txt"
python tools/critic_stream_probe.py f64 2>/dev/null | grep -v amdgpu > "$O/${R}_critic_stream_probe_f64.txt"
python tools/split_probe.py 2>/dev/null | grep "S =" > "$O/${R}_split_probe.txt"
python tools/critic_fit_probe.py quadratic 2>/dev/null | grep -v amdgpu > "$O/${R}_critic_fit_probe.txt"
[ -x build/event_probe ] && ./build/event_probe > "$O/${R}_event_probe.txt" 2>&1
[ -f rcognita_amd/lib/librcg_dev.so ] && python tools/packed_sweep.py > "$O/${R}_packed_sweep.txt" 2>&1
python -m pytest tests/test_hip_teacher_forced.py tests/test_hip_ref_traces.py tests/test_hip_configs.py -q -s -k "F7 or free_running or trace or teacher" 2>&1 | grep -E "^TRACE|^TEACHER|FREE RUN|passed|failed" > "$O/${R}_trace_and_free_run_tests.txt"
python tools/valu_probe.py search 2>/dev/null | tail -1 > "$O/${R}_search_probe.json"
python tools/fill_bw_probe.py 2>/dev/null > "$O/${R}_fill_bw_probe.txt"
# round 4: T ticks per launch against per-tick launches (small batches), the optimiser's cost against SLSQP's, the search
python -m pytest tests/test_hip_ticks.py -q -s -k "rate or launch_bound or persistent" 2>&1 | grep -E "B=|persistent|passed|failed" > "$O/${R}_ticks_rates.txt"
python -m pytest tests/test_hip_optimizer.py tests/test_hip_search.py -q -s 2>&1 | grep -E "SLSQP|slsqp|search|gap|twin|passed|failed" > "$O/${R}_optimizer_and_search_quality.txt"
if [ -f rcognita_amd/lib/librcg_dev.so ]; then
  MODE=mpc bash tools/ab_min_k.sh > "$O/${R}_ab_min_k.txt" 2>&1
  MODE=crit bash tools/ab_min_k.sh 2>&1 | grep -E "==|AB" > "$O/${R}_ab_packed_critic.txt"
fi
fi
ls "$O" | wc -l
