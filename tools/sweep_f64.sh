# A/B of k_actor_dma<double>'s residency (dev library knobs; interleaved on one device): the shapes whose residency the
# element-based thresholds of round 6 changed (default = 4 blocks per CU now) against the former 2.   bash tools/sweep_f64.sh
export SWEEP_DTYPE=f64
for shape in "SWEEP_N=5" "SWEEP_N=7" "SWEEP_K=64" "SWEEP_K=96" "SWEEP_B=16384" "SWEEP_B=32768" "SWEEP_N=10"; do for rep in 1 2; do
env $shape RCG_PER_CU=2 python tools/knob_sweep.py 2>/dev/null | tail -1
env $shape python tools/knob_sweep.py 2>/dev/null | tail -1
done; done
