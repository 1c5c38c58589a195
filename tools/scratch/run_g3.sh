mkdir -p gpurun_out/g2
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "graph_mode" > gpurun_out/g2/test1.log 2>&1; echo "alone rc $?"; tail -4 gpurun_out/g2/test1.log | cut -c1-300
python tools/scratch/graph_try.py 8 opt 2>&1 | tail -4
