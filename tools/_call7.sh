cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_hip_ticks.py tests/test_hip_knobs.py tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_critic.py -m gpu -x -q > gpurun_out/t_gpu_sub.txt 2>&1; tail -3 gpurun_out/t_gpu_sub.txt
python bench.py --regime generated --no-cpu-baseline --no-secondary > gpurun_out/bench_gen2.json 2>/dev/null; python -c "
import json;b=json.load(open('gpurun_out/bench_gen2.json'));print('generated', b['value'], b['ms_per_step'])"
python tools/critic_stream_probe.py f32 2>/dev/null | grep -v amdgpu > gpurun_out/critic_stream_probe_f32_b.txt; cat gpurun_out/critic_stream_probe_f32_b.txt
python tools/critic_stream_probe.py f64 2>/dev/null | grep -v amdgpu > gpurun_out/critic_stream_probe_f64_b.txt
SQ="SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
rm -rf gpurun_out/prof_valu
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/prof_valu -o v -- \
  python3 tools/valu_probe.py main > gpurun_out/valu_units.json 2> gpurun_out/prof_valu.log
tail -c 200 gpurun_out/valu_units.json
