"""`PointNet2SSG`: a PointNet++ single-scale-grouping encoder for the siamese ReID matcher, composed
from the reference's mmdet3d-style `PointSAModule` semantics (D-FPS -> ball query -> shared MLP ->
max pool; mmdet3d/ops/pointnet_modules/point_sa_module.py:302-354).

The reference ships the ops but no ReID config that uses them; BASELINE
config 2 ("PointNet++ SSG siamese, 1024 pts") is therefore this build's own composition, fixed in
SURVEY.md 8d:  SA(512, r=0.2, K=32, [3 -> 64, 64, 128]) -> SA(128, r=0.4, K=64, [131 -> 128, 128, 256])
-> Conv1d(256 -> 64).  It returns the 128 abstracted points and their 64-d features, which the
unchanged matching head (cross attention over xyz + features) consumes.  Registered in ReIDNet's
module_obj as 'PointNet2SSG'."""
import torch.nn as nn

from pcr_amd import engine


class PointNet2SSG(nn.Module):
    def __init__(self, num_points=(512, 128), radii=(0.2, 0.4), num_samples=(32, 64),
                 sa_channels=((64, 64, 128), (128, 128, 256)), conv_out=64, in_channels=0):
        super().__init__()
        from mmdet3d.ops.pointnet_modules import PointSAModule
        self.SA_modules = nn.ModuleList()
        last = in_channels
        for npnt, r, k, ch in zip(num_points, radii, num_samples, sa_channels):
            self.SA_modules.append(PointSAModule(mlp_channels=[last] + list(ch), num_point=npnt, radius=r,
                                                 num_sample=k, use_xyz=True))
            last = ch[-1]
        self.cov_final = nn.Conv1d(last, conv_out, kernel_size=1)

    def forward(self, pointcloud, numpoints=None):
        """pointcloud (B,N,3[+C]) -> (xyz (B,M,3), features (B,conv_out,M)) for the M abstracted points"""
        xyz = pointcloud[..., 0:3].contiguous()
        feats = pointcloud[..., 3:].transpose(1, 2).contiguous() if pointcloud.size(-1) > 3 else None
        for sa in self.SA_modules:
            xyz, feats, _ = sa(xyz, feats)
        key = (str(feats.device), engine.param_version(self.cov_final))
        if getattr(self, "_cf_key", None) != key:
            object.__setattr__(self, "_cf_w", engine.pack_weight_dual(self.cov_final.weight, feats.device))
            object.__setattr__(self, "_cf_b", self.cov_final.bias.detach().to(feats.device).float().contiguous())
            object.__setattr__(self, "_cf_key", key)
        return xyz, engine.dense(feats, self._cf_w, self.cov_final.weight.shape[0], None, self._cf_b, 0)
