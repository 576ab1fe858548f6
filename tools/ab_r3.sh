for L in reart_amd/csrc/libreart_hip_q768.so reart_amd/csrc/libreart_hip.so reart_amd/csrc/libreart_hip_q768.so reart_amd/csrc/libreart_hip.so; do
  REART_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-tail --no-secondary --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', 'single', d['value'], d['roofline']['kernel_ms'], 'sweep', d['sweep']['value'])"
done
python -m pytest tests -m gpu -q 2>&1 | tail -2
