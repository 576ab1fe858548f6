source tools/ab.sh
run X=1
run REART_LIB=reart_amd/csrc/libreart_hip_w7.so
run REART_LIB=reart_amd/csrc/libreart_hip_w8.so
run X=1
run REART_LIB=reart_amd/csrc/libreart_hip_w7.so
run REART_LIB=reart_amd/csrc/libreart_hip_w8.so
