for env in "X=1" "REART_SPARSE=20" "REART_SPARSE=64" "REART_SHARE=0" "REART_FWD_PTS=64" "REART_BWD_PTS=64"; do
  env $env timeout 300 python bench.py --no-cpu-baseline --no-tail --no-secondary --profile-steps 0 --sweep-instances 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$env', 'single', d['value'], 'sweep', d['sweep']['value'])"
done
