source tools/ab.sh
python -m pytest tests/test_warm_gpu.py tests/test_step_gpu.py tests/test_parity2_gpu.py tests/test_knn_gpu.py tests/test_sweep_gpu.py tests/test_gumbel_stream_gpu.py -m gpu -q -x 2>&1 | tail -5
run REART_LIB=reart_amd/csrc/libreart_hip_base.so
run X=1
run REART_LIB=reart_amd/csrc/libreart_hip_base.so
run X=1
