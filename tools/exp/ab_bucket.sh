# epoch-mode (--shuffle) step time against the shape-bucket granularity
for b in "64 1280" "32 640" "16 320" "8 160"; do
  python bench.py --shuffle --bucket $b --no-secondary --no-cpu-baseline --steps 300 --warmup 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('bucket', r['config']['bucket'], r['ms_per_step'], r['value'], r['slots'])"
done
