mkdir -p gpurun_out/r3t
python -m pytest tests -x -q -m gpu -k "train or model or distributed or abi or fuzz" > gpurun_out/r3t/test.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r3t/test.log
python tools/train_detail.py > gpurun_out/r3t/train_detail.log 2>&1; head -30 gpurun_out/r3t/train_detail.log | cut -c1-110
