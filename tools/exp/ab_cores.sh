B=$PWD/dostransformer_amd/csrc/build
run() { python bench.py --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['roofline']['avg_us'])"; }
for i in 1 2 3; do
  unset DOSX_LIB; unset DOSX_WGRAD_NB; run base
  export DOSX_LIB=$B/libdosx_cores.so; export DOSX_WGRAD_NB=2; run "ffn168+wgrad80+nb2"
  export DOSX_LIB=$B/libdosx_cores.so; unset DOSX_WGRAD_NB; run "ffn168+wgrad80+nb3"
  export DOSX_LIB=$B/libdosx_ffn3.so; export DOSX_WGRAD_NB=2; run "ffn168+wgrad93+nb2"
done
