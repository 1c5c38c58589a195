_base_ = ["../_base_/model.py", "../_base_/runtime.py"]
model = dict(eval_only=True, backbone=dict(nsample=[48, 64, 64]), local_stage1=dict(type="x"))
data = dict(val=dict(subsample_sparse=256))
