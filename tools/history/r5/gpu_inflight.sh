for m in "" "--mode path"; do for k in 2 3 4; do
python bench.py $m --frames-in-flight $k --no-cpu-baseline --no-extras --steps 1500 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$m in flight $k:', round(d['value']), round(d['ms_per_step']*1e3,2))"
done; done
