# A/B of two builds of libdosx on one configuration: ab_lib.sh <config> <steps> <lib B> [site substring to print]
cfg=$1; steps=$2; libb=$3; site=${4:-N512,K256}
for lib in "" "$libb" "" "$libb"; do
  DOSX_LIB=$lib python bench.py --config $cfg --no-secondary --no-cpu-baseline --steps $steps --warmup 10 --kernels-out /tmp/k.json 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=json.load(open('/tmp/k.json'))
s=[x for x in k['sites'] if '$site' in x['site']]
print('lib', '$lib' or 'default', r['ms_per_step'], ' | '.join('%s %.1f us' % (x['site'][5:40], x['avg_us']) for x in s))"
done
