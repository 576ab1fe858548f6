for k in 2 3 4 6; do for s in 1 2 3; do
  REART_PRUNE_SPLIT=$s REART_PRUNE_SPLIT3=$s timeout 300 python bench.py --no-cpu-baseline --no-tail --no-secondary --profile-steps 0 --sweep-instances $k --steps 600 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('K', d['sweep']['instances_per_gpu'], 'S', $s, 'sweep', d['sweep']['value'])"
done; done
