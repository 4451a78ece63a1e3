for i in 1 2 3; do
for f in "--sync-loop" ""; do python bench.py --no-cpu-baseline --no-exact-leg $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', round(d['value'],1), round(d['ms_per_step'],2))"; done; done
