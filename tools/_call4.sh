cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=rcognita_amd/lib/librcg_dev.so
{ echo "== SQL quad-lin C2: A default (2 blocks/CU), B RCG_PER_CU=4"; python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 3 sql
  echo "== SQL quad-lin C2: B RCG_PER_CU=8"; python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=8 --rounds 2 sql
  echo "== RQL quad-lin C2: B RCG_PER_CU=4"; AB_MODE=RQL AB_CS=quad-lin python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 2 stream
  echo "== SQL quadratic C2: B RCG_PER_CU=4"; AB_MODE=SQL AB_CS=quadratic python tools/ab_lib.py --a $D --b $D --b-env RCG_PER_CU=4 --rounds 2 stream
} > gpurun_out/ab_sql_per_cu.txt 2>&1
grep -E "==|AB" gpurun_out/ab_sql_per_cu.txt
PART=2 bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04_p2.log 2>&1
tail -2 gpurun_out/profile_round_r04_p2.log
