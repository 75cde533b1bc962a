export SWEEP_DTYPE=f64
for rep in 1 2; do
for g in 4 8 16 32; do RCG_GPW=$g python tools/knob_sweep.py 2>/dev/null | tail -1; done
for p in 2 4 8; do RCG_PER_CU=$p python tools/knob_sweep.py 2>/dev/null | tail -1; done
python tools/knob_sweep.py 2>/dev/null | tail -1
done
