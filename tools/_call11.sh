cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=rcognita_amd/lib/librcg_dev.so
for b in 8192 16384 21846 32768; do
  echo "== gen B=$b: A default, B RCG_GPW=8"; AB_B=$b python tools/ab_lib.py --a $D --b $D --b-env RCG_GPW=8 --rounds 2 gen 2>&1 | grep -E "AB|FAILED"
  echo "== gen B=$b: A RCG_GPW=4, B RCG_NO_TICK_FUSE=1"; AB_B=$b python tools/ab_lib.py --a $D --b $D --a-env RCG_GPW=4 --b-env RCG_NO_TICK_FUSE=1 --rounds 2 gen 2>&1 | grep -E "AB|FAILED"
done
