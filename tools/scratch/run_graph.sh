python tools/scratch/graph_try.py 256 2>&1 | tail -6
python tools/scratch/graph_try.py 256 opt 2>&1 | tail -6
python tools/scratch/graph_try.py 16 opt 2>&1 | tail -4
