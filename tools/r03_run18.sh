for mb in 768 576 384 192; do
CNIIC_KM_MAX_BLOCKS=$mb timeout -k 10 300 python bench.py --cpu-sample 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$mb', d['ms_per_step'], r['frac'], {k:v['us'] for k,v in r['by_class'].items()})"
done
