# (needs the DOSX_EDGE_ENC_EARLY knob of the experiment) A/B: edge encoder backward right after dL/de_0 exists (main stream, its weight gradients in layer 0's group) vs at the tail
for v in 1 0 1 0 1 0; do
  export DOSX_EDGE_ENC_EARLY=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('early', os.environ['DOSX_EDGE_ENC_EARLY'], r['ms_per_step'])"
done
for v in 1 0 1 0; do
  export DOSX_EDGE_ENC_EARLY=$v
  python bench.py --config edos_h256_b64 --no-secondary --no-cpu-baseline --steps 60 --warmup 16 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos early', os.environ['DOSX_EDGE_ENC_EARLY'], r['ms_per_step'])"
done
