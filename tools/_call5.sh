cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PROBE_LIB=rcognita_amd/lib/librcg_dev.so
for rep in 1 2; do
for dt in f32 f64; do
  python tools/critic_stream_probe.py $dt 65536 all 2>/dev/null | grep -v amdgpu | sed "s/^/percu_default rep$rep /" >> gpurun_out/per_cu_matrix.txt
  RCG_PER_CU=4 python tools/critic_stream_probe.py $dt 65536 all 2>/dev/null | grep -v amdgpu | sed "s/^/percu_4 rep$rep /" >> gpurun_out/per_cu_matrix.txt
done
done
wc -l gpurun_out/per_cu_matrix.txt
