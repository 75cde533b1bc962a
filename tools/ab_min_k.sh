# A/B of how few candidates per env the DMA kernels serve (dev build knobs), one GPU, interleaved child processes:
#   MODE=mpc   K = 33 .. 39: k_actor_dma's one ragged tile (A) against k_actor (B: RCG_DMA_MINK=40)
#   MODE=crit  RQL / SQL, K = 8 .. 32: k_actor_dma_packed / k_actor_dma (A) against k_actor (B: RCG_NO_PACK=1 RCG_DMA_MINK=64)
set -e
D=rcognita_amd/lib/librcg_dev.so
if [ "${MODE:-crit}" = mpc ]; then
  for k in 33 36 39; do
    echo "== MPC K=$k"; AB_K=$k python tools/ab_lib.py --a $D --b $D --b-env RCG_DMA_MINK=40 --rounds 3 stream
  done
else
  for k in ${KS:-8 16 24 32}; do
    echo "== RQL quad-nomix K=$k"
    AB_K=$k AB_MODE=RQL python tools/ab_lib.py --a $D --b $D --b-env RCG_NO_PACK=1,RCG_DMA_MINK=64 --rounds 3 stream
    echo "== SQL quad-lin K=$k"
    AB_K=$k AB_MODE=SQL AB_CS=quad-lin python tools/ab_lib.py --a $D --b $D --b-env RCG_NO_PACK=1,RCG_DMA_MINK=64 --rounds 3 stream
  done
fi
