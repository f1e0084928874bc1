for w in 5 16 40 5 16 40; do
  python3 bench.py --gpus 1 --steps 20 --warmup $w --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('warmup', r['warmup'], r['ms_per_step'])"
done
