export MOPTIX_DEVICE_LIB=libmoptix_ev.so
EVLOG=$PWD/gpurun_out/r06_evlog_share.bin timeout 300 python tools/gpu_lone_path.py 2>&1 | tail -2
python tools/evlog_timeline.py gpurun_out/r06_evlog_share.bin 150 2
