source tools/ab.sh
python -m pytest tests/test_warm_gpu.py tests/test_step_gpu.py tests/test_parity2_gpu.py tests/test_sweep_gpu.py tests/test_gumbel_stream_gpu.py -m gpu -q -x 2>&1 | tail -5
run REART_SHARE=3 X=1
run X=1
run REART_SPARSE=20 X=1
run REART_SPARSE=64 X=1
run REART_PRUNE_SPLIT=2 REART_PRUNE_SPLIT3=2
run REART_PRUNE_SPLIT=4 REART_PRUNE_SPLIT3=4
