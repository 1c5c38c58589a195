_base_ = ["../_base_/model.py", "./dup_model.py"]
