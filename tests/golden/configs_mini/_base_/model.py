# own test fixture (not a reference file): exercises the _base_ merge rules of pcr_amd.config
width = 64
hidden = width * 2
model = dict(
    type="ReIDNet",
    backbone_list=[128, 64, 32],
    backbone=dict(type="Pointnet_Backbone", input_channels=0, conv_out=width, nsample=[32, 48, 48]),
    local_stage1=dict(),
    heads=[dict(type="LinearRes", n_in=hidden, n_out=hidden), dict(type="Linear", in_features=hidden, out_features=1)],
)
