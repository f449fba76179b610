cd /root/repo
KRYST_ILU_WAVE=2 KRYST_HIP_LIB=kryst_amd/lib/libkryst_hip_trace.so timeout -k 10 100 python3 tools/tq_trace.py 256 0,0 1,1 5,5 6,5 6,6 2>&1 | tail -24
