set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t11.log 2>&1; tail -12 gpurun_out/r03_t11.log
