set -e
mkdir -p gpurun_out
python tools/philox_probe.py > gpurun_out/philox_probe.txt 2>&1 || (tail -20 gpurun_out/philox_probe.txt; exit 1)
tail -40 gpurun_out/philox_probe.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r03_base.txt 2>&1 || (tail -30 gpurun_out/gputest_r03_base.txt; exit 1)
tail -5 gpurun_out/gputest_r03_base.txt
