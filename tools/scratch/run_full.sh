mkdir -p gpurun_out/full
python -m pytest tests -x -q -m gpu > gpurun_out/full/test.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/full/test.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/full/smoke.log
