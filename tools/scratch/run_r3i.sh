mkdir -p gpurun_out/r3i
python -m pytest tests/test_gpu_train_variants.py -q -s -m gpu > gpurun_out/r3i/new.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r3i/new.log | cut -c1-600 | tail -12
grep -o '{"tag": "dgcnn".*}' gpurun_out/r3i/new.log | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('loss',d['loss'],d['ref_loss'],'gn',d['grad_norm'],d['ref_grad_norm'])
    for k in d['vs_ref32']: print('  %-40s ours-vs-32 %.1e  ref32-vs-64 %.1e  ours-vs-64 %.1e'%(k,d['vs_ref32'][k],d['ref32_vs_ref64'][k],d['vs_ref64'][k]))
"
