# A/B (needs the DOSX_SIDE_NODE_ENC knob of the experiment): node encoder on the side stream (1) vs in line (0)
for v in 1 0 1 0 1 0; do
  export DOSX_SIDE_NODE_ENC=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('side', os.environ['DOSX_SIDE_NODE_ENC'], r['ms_per_step'])"
done
for v in 1 0 1 0; do
  export DOSX_SIDE_NODE_ENC=$v
  python bench.py --config edos_h256_b64 --no-secondary --no-cpu-baseline --steps 60 --warmup 16 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos side', os.environ['DOSX_SIDE_NODE_ENC'], r['ms_per_step'])"
done
