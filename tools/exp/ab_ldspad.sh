# (needs the DOSX_WGRAD_LDS_PAD knob of the experiment: extra dynamic LDS bytes on the wgrad_grouped_kernel launch in dosx_grad_flush)
# A/B: one weight-gradient workgroup per CU (LDS pad) instead of two
for v in 0 8192 0 8192; do
  export DOSX_WGRAD_LDS_PAD=$v
  python bench.py --config edos_h256_b64 --no-secondary --no-cpu-baseline --steps 60 --warmup 16 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos pad', os.environ['DOSX_WGRAD_LDS_PAD'], r['ms_per_step'], r['roofline']['avg_us'])"
done
for v in 0 32768 0 32768 0 32768; do
  export DOSX_WGRAD_LDS_PAD=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('cfg2 pad', os.environ['DOSX_WGRAD_LDS_PAD'], r['ms_per_step'], r['roofline']['avg_us'])"
done
