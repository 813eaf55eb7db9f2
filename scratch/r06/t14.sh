cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('value', round(d['value'],1), 'roofline frac', round(d['roofline']['frac'],3), 'traffic', d['roofline']['traffic'], 'us', round(d['roofline']['us_per_launch'],2))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['unit'], d['cpu_baseline']['cores'])
print({k:(v.get('tok_s') if isinstance(v,dict) else v) for k,v in d['extra'].items()})
"
