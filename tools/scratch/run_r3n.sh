mkdir -p gpurun_out/r3n
python tools/train_aten_profile.py > gpurun_out/r3n/aten_prof.log 2>&1; echo "aten rc $?"
python tools/train_detail.py > gpurun_out/r3n/train_detail.log 2>&1; echo "detail rc $?"
python bench.py --workload pt128_train --no-cpu-baseline > gpurun_out/r3n/pt128_train.json 2> gpurun_out/r3n/pt128_train.err; echo "bench rc $?"
tail -c 600 gpurun_out/r3n/pt128_train.json
