python -m pytest tests/test_gpu_conv.py tests/test_gpu_engine.py -x -q 2>&1 | tail -n 2
FRX_LIB=scripts/_ab/libfrx_new.so python scripts/layer_times.py > gpurun_out/lt_new.txt 2>&1
FRX_LIB=scripts/_ab/libfrx_new.so python scripts/layer_times.py > gpurun_out/lt_new2.txt 2>&1
python scripts/lt_compare.py gpurun_out/lt_new.txt gpurun_out/lt_new2.txt | tail -n 26
bash scripts/ab.sh > /dev/null 2>&1; cat gpurun_out/ab.log
