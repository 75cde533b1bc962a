cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ev_r04; mkdir -p $O
python -m pytest tests/test_hip_search.py tests/test_hip_ticks.py tests/test_hip_configs.py tests/test_hip_knobs.py -m gpu -x -q > gpurun_out/t_sub2.txt 2>&1 || { tail -30 gpurun_out/t_sub2.txt; exit 1; }
tail -1 gpurun_out/t_sub2.txt
python bench.py > $O/r04_bench.json 2> $O/err.log
python bench.py --config C5 --no-cpu-baseline --no-secondary > $O/r04_bench_c5.json 2>> $O/err.log
python bench.py --config C5 --steps 10000 --warmup 100 --no-cpu-baseline --no-secondary > $O/soak_C5_f32.json 2>> $O/err.log
python bench.py --regime generated --no-cpu-baseline --no-secondary > $O/r04_bench_c2_generated.json 2>> $O/err.log
python - <<'PY'
import json
for f in ("r04_bench","r04_bench_c5","soak_C5_f32","r04_bench_c2_generated"):
    b=json.loads(open(f"gpurun_out/ev_r04/{f}.json").read().strip().split("\n")[-1])
    print(f, "%.4g"%b["value"], b["ms_per_step"], (b.get("parity") or {}).get("ok"))
b=json.loads(open("gpurun_out/ev_r04/r04_bench.json").read().strip().split("\n")[-1])
print(json.dumps(b["secondary"]["produced_stream"])[:300])
PY
