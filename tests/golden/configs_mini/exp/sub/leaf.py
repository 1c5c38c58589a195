_base_ = "../base_exp.py"
model = dict(backbone_list=[1024, 512, 256], heads=[dict(type="Linear", in_features=8, out_features=1)],
             local_stage1=dict(_delete_=True, other=1))
tags = ["a", "b"]
