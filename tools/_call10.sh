cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=rcognita_amd/lib/librcg_dev.so
python tools/ab_lib.py --a $D --b $D --b-env RCG_GPW=8 --rounds 5 pool 2>&1 | grep -E "AB|FAILED"
python tools/ab_lib.py --a $D --b $D --a-env RCG_GPW=4 --b-env RCG_NO_TICK_FUSE=1 --rounds 5 pool 2>&1 | grep -E "AB|FAILED"
python tools/ab_lib.py --a $D --b $D --a-env RCG_GPW=1 --b-env RCG_NO_PK=1 --rounds 5 pool 2>&1 | grep -E "AB|FAILED"
