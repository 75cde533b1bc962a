cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
PART=1 bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04_p1.log 2>&1; tail -2 gpurun_out/profile_round_r04_p1.log | cut -c1-300
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/ev_r04/r04_driver_form_$i.json 2>> gpurun_out/ev_r04/err.log; done
python bench.py --config C3 --no-cpu-baseline --no-secondary > gpurun_out/ev_r04/r04_bench_c3_two_handles.json 2>> gpurun_out/ev_r04/err.log
python bench.py --config C5 --no-cpu-baseline --no-secondary > gpurun_out/ev_r04/r04_bench_c5.json 2>> gpurun_out/ev_r04/err.log
python bench.py --config C3 --regime generated --no-cpu-baseline --no-secondary > gpurun_out/ev_r04/r04_bench_c3_generated.json 2>> gpurun_out/ev_r04/err.log
ls gpurun_out/ev_r04 | wc -l
