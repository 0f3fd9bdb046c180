set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_coarse.py -x -q > gpurun_out/r03_t5.log 2>&1; tail -15 gpurun_out/r03_t5.log
