cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu_full.txt 2>&1; tail -4 gpurun_out/t_gpu_full.txt
D=rcognita_amd/lib/librcg_dev.so
{ echo "== NI MPC C2 shape: B RCG_PER_CU=4"; AB_SYS=3wrobotNI python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 2 stream
  echo "== NI MPC N=20: B RCG_PER_CU=4"; AB_SYS=3wrobotNI AB_N=20 python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 2 stream
  echo "== 2tank N=20 B=131072 RQL quadratic (C3): B RCG_PER_CU=4"; AB_SYS=2tank AB_N=20 AB_B=131072 AB_MODE=RQL AB_CS=quadratic python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 2 stream
  echo "== 2tank N=20 B=131072 MPC: B RCG_PER_CU=4"; AB_SYS=2tank AB_N=20 AB_B=131072 python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 2 stream
  echo "== 3wrobot MPC N=20: B RCG_PER_CU=4"; AB_N=20 python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 2 stream
  echo "== 3wrobot MPC C2: B RCG_PER_CU=4"; python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 2 stream
} > gpurun_out/ab_per_cu_more.txt 2>&1
grep -E "==|AB" gpurun_out/ab_per_cu_more.txt
