for a in "" "--columns 2" "--columns 2 --columns-min-width 900"; do echo "== $a"; timeout -k 10 120 python bench.py --no-cpu-baseline --no-ceiling --no-photo $a 2>&1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%7.0f images/s   pyramid %6.2f us/image   frac %.3f' % (d['value'], d['kernels_ms_per_image']['pyramid']*1e3, d['roofline']['frac']))"; done
