cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ev_r04; mkdir -p $O
python -m pytest tests/test_hip_search.py -m gpu -x -q > gpurun_out/t_search.txt 2>&1 || { tail -20 gpurun_out/t_search.txt; exit 1; }
tail -1 gpurun_out/t_search.txt
python bench.py > $O/r04_bench.json 2> $O/err.log
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/r04_driver_form_$i.json 2>> $O/err.log; done
python bench.py --config C3 --no-cpu-baseline --no-secondary > $O/r04_bench_c3_two_handles.json 2>> $O/err.log
python bench.py --config C3 --parts 1 --no-cpu-baseline --no-secondary > $O/r04_bench_c3_one_handle.json 2>> $O/err.log
python bench.py --config C5 --no-cpu-baseline --no-secondary > $O/r04_bench_c5.json 2>> $O/err.log
python bench.py --config C3 --regime generated --no-cpu-baseline --no-secondary > $O/r04_bench_c3_generated.json 2>> $O/err.log
# soak: long runs of the bench workloads (frozen envs counted, parity of the run's own outputs)
for spec in "C2 20000 f32" "C3 6000 f32" "C5 10000 f32" "C3 3000 f64"; do
  set -- $spec
  python bench.py --config $1 --steps $2 --warmup 100 --dtype $3 --no-cpu-baseline --no-secondary > $O/soak_$1_$3.json 2>> $O/err.log
  python - "$O/soak_$1_$3.json" "$1" "$3" "$2" >> $O/r04_soak.txt <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=b["roofline"]; t=b.get("returns_summary") or r.get("returns_summary") or {}
print(sys.argv[2], sys.argv[3], "steps", sys.argv[4], "value %.3e"%b["value"], "ms %.4f"%b["ms_per_step"], "frac %.3f"%(r.get("frac") or 0), "n_failed", t.get("n_failed"), "count", t.get("count"), "parity", (b.get("parity") or {}).get("ok"))
PY
done
cat $O/r04_soak.txt
