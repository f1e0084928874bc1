# A/B: M-splits of the encoder weight gradients that run alone at the end of the step
for v in 32 64 0 32 64 0; do
  export DOSX_WGRAD_TAIL_SPLITS=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('tail splits', os.environ['DOSX_WGRAD_TAIL_SPLITS'], r['ms_per_step'])"
done
for v in 32 64 0 32 64 0; do
  export DOSX_WGRAD_TAIL_SPLITS=$v
  python bench.py --config edos_h256_b64 --no-secondary --no-cpu-baseline --steps 60 --warmup 16 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos tail splits', os.environ['DOSX_WGRAD_TAIL_SPLITS'], r['ms_per_step'])"
done
