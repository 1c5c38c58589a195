"""`Pointnet_Backbone`: the reference's "Point-Transformer" encoder (mmdet3d/models/
backbone_net.py:25-124): 3 edge set-abstraction layers with linear self-attention, 3 linear
cross-attention feature-propagation layers back to all N points, 1x1 conv head."""
import torch.nn as nn

from .pointnet2_utils import PointNetFeaturePropagationSA, PointNetSetAbstractionEdgeSA


_NO_KNN2 = bool(__import__("os").environ.get("PCR_NO_KNN2"))   # diagnostics: one neighbour search per level


class Pointnet_Backbone(nn.Module):
    def __init__(self, input_channels=3, use_xyz=True, conv_out=32, mul=1, radius=[0.3, 0.5, 0.7],
                 nsample=[32, 48, 48]):
        super().__init__()
        sa1, sa2, sa3 = 32 * mul, 64 * mul, 128 * mul
        self.SA_modules = nn.ModuleList()
        for r, k, widths in ((radius[0], nsample[0], [input_channels, sa1, sa1, sa1]),
                             (radius[1], nsample[1], [sa2, sa2, sa2, sa2]),
                             (radius[2], nsample[2], [sa3, sa3, sa3, sa3])):
            self.SA_modules.append(PointNetSetAbstractionEdgeSA(npoint=None, radius=r, nsample=k, mlp=widths,
                                                                sampling="RANDOM", use_xyz=use_xyz, use_knn=True))
        self.FP_modules = nn.ModuleList()
        self.FP_modules.append(PointNetFeaturePropagationSA(mlp=[67, sa1, sa1], mlp_inte=[sa2, 3, sa2, sa2, sa1]))
        self.FP_modules.append(PointNetFeaturePropagationSA(mlp=[160, sa3, sa2], mlp_inte=[sa3, sa1, sa3, sa2, sa2]))
        self.FP_modules.append(PointNetFeaturePropagationSA(mlp=[192, sa3, sa3], mlp_inte=[sa3, sa2, sa3, sa2, sa3]))
        self.cov_final = nn.Conv1d(sa1, conv_out, kernel_size=1)
        # cov_final is evaluated inside the last FP launch
        self.FP_modules[0].interpolation.fuse_final_conv(self.cov_final)

    def _break_up_pc(self, pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def forward(self, pointcloud, numpoints):
        """pointcloud (B,N,3+C) -> (xyz (B,N,3), features (B,conv_out,N))"""
        if self.training:
            # differentiable graph with BatchNorm batch statistics (pcr_amd/train_graph.py); eval mode below is
            # the fused HIP path
            from pcr_amd import train_graph
            return train_graph.backbone(self, pointcloud, numpoints)
        xyz, features = self._break_up_pc(pointcloud)
        l_xyz, l_features = [xyz], [features]
        shared = None       # the second level's neighbours, when the first level's search ranked them too
        for i, sa in enumerate(self.SA_modules):
            knn_idx = None
            if i == 0 and len(self.SA_modules) > 1 and hasattr(sa, "shares_knn_with") and not _NO_KNN2 and \
                    sa.shares_knn_with(self.SA_modules[1], xyz.shape[1], numpoints[0], numpoints[1]):
                from pcr_amd import engine
                knn_idx, shared = engine.knn_prefix2(xyz.contiguous(), numpoints[0], sa.nsample, numpoints[1],
                                                     self.SA_modules[1].nsample)
            elif i == 1 and shared is not None:
                knn_idx = shared
            li_xyz, li_features = sa(l_xyz[i], l_features[i], numpoints[i]) if knn_idx is None else \
                sa(l_xyz[i], l_features[i], numpoints[i], knn_idx=knn_idx)
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        l_features[0] = xyz.transpose(1, 2).contiguous()
        for i in (2, 1, 0):
            l_features[i] = self.FP_modules[i](l_xyz[i], l_xyz[i + 1], l_features[i], l_features[i + 1])
        if self.FP_modules[0].interpolation._final is not None:
            return l_xyz[0], l_features[0]      # cov_final already applied inside the FP_modules[0] launch
        return l_xyz[0], self._cov_final_unfused(l_features[0])

    def _cov_final_unfused(self, x):
        from pcr_amd import engine
        key = (str(x.device), engine.param_version(self.cov_final))
        if getattr(self, "_cf_key", None) != key:
            object.__setattr__(self, "_cf_w", engine.pack_weight(self.cov_final.weight, x.device))
            object.__setattr__(self, "_cf_b", self.cov_final.bias.detach().to(x.device).float().contiguous())
            object.__setattr__(self, "_cf_key", key)
        return engine.dense(x.contiguous(), self._cf_w, self.cov_final.weight.shape[0], None, self._cf_b, 0)
