# k_actor_dma<double> taken apart on the C2 shape (dev library, RCG_DBG bits: 1 no rollout, 2 no argmin / writes, 4 no env-state loads),
# interleaved on one device; the same for float for comparison.   bash tools/sweep_f64.sh
for dt in f64 f32; do export SWEEP_DTYPE=$dt
for rep in 1 2 3; do
python tools/knob_sweep.py 2>/dev/null | tail -1
RCG_DBG=1 python tools/knob_sweep.py 2>/dev/null | tail -1
RCG_DBG=7 python tools/knob_sweep.py 2>/dev/null | tail -1
done; done
