import os
seed = 66
data = dict(samples_per_gpu=256, val=dict(subsample_sparse=128, path="{}/val".format("root")))
def helper():
    return 1
