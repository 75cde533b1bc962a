cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_hip_search.py -m gpu -x -q > gpurun_out/t_search.txt 2>&1 || { tail -20 gpurun_out/t_search.txt; exit 1; }
tail -2 gpurun_out/t_search.txt
SQ="SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
rm -rf gpurun_out/prof_valu
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/prof_valu -o v -- \
  python3 tools/valu_probe.py main > gpurun_out/valu_units.json 2> gpurun_out/prof_valu.log
tail -c 300 gpurun_out/valu_units.json
PART=2 bash tools/evidence_round.sh r04 > gpurun_out/evidence_r04_p2.log 2>&1
tail -3 gpurun_out/evidence_r04_p2.log
