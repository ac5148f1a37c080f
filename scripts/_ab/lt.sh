python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -n 2
FRX_LIB=scripts/_ab/libfrx_old.so python scripts/layer_times.py > gpurun_out/lt_old.txt 2>&1
FRX_LIB=scripts/_ab/libfrx_new.so python scripts/layer_times.py > gpurun_out/lt_new.txt 2>&1
FRX_LIB=scripts/_ab/libfrx_old.so python scripts/layer_times.py > gpurun_out/lt_old2.txt 2>&1
FRX_LIB=scripts/_ab/libfrx_new.so python scripts/layer_times.py > gpurun_out/lt_new2.txt 2>&1
python scripts/lt_compare.py gpurun_out/lt_old.txt gpurun_out/lt_new.txt gpurun_out/lt_old2.txt gpurun_out/lt_new2.txt
