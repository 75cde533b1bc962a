cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu_full.txt 2>&1; tail -2 gpurun_out/t_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
SQ="SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
rm -rf gpurun_out/prof_valu_pool
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/prof_valu_pool -o v -- \
  python3 tools/valu_probe.py pool > gpurun_out/valu_units_pool.json 2> gpurun_out/prof_valu_pool.log
tail -c 200 gpurun_out/valu_units_pool.json
rm -rf gpurun_out/prof_kt_configs
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt_configs -o kt -- \
  python3 tools/bench_configs.py > gpurun_out/bench_configs_r04_profiled.json 2> gpurun_out/prof_kt_configs.log
python tools/bench_configs.py > gpurun_out/bench_configs_r04.json 2> gpurun_out/bench_configs.err
tail -c 300 gpurun_out/bench_configs_r04.json
