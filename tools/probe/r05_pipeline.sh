cd $GRAFT_REPO_ROOT
for p in 1 2 3; do
python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sub --verify 4 --pipeline $p 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pipeline $p value %.0f ms/step %.4f kernel %.4f serial %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('serial_step',{}).get('ms_per_step')))"
done
