"""DGCNN backbone returning per-point features without global pooling (reference:
mmdet3d/models/dgcnn_orig.py -- knn :22-29, get_graph_feature :32-56, DGCNN :89-152).  Parameter names match the
reference, including its double registration of every BatchNorm (self.bn{i} is also conv{i}.1), so reference
checkpoints load unchanged."""
import torch.nn as nn


class DGCNN(nn.Module):
    def __init__(self, dropout=0.5, emb_dims=1024, k=20, output_channels=40):
        super().__init__()
        self.k = k
        self.bn1 = nn.BatchNorm2d(64)
        self.bn2 = nn.BatchNorm2d(64)
        self.bn3 = nn.BatchNorm2d(128)
        self.bn4 = nn.BatchNorm2d(256)
        self.bn5 = nn.BatchNorm1d(emb_dims)
        act = lambda: nn.LeakyReLU(negative_slope=0.2)   # noqa: E731
        self.conv1 = nn.Sequential(nn.Conv2d(6, 64, kernel_size=1, bias=False), self.bn1, act())
        self.conv2 = nn.Sequential(nn.Conv2d(64 * 2, 64, kernel_size=1, bias=False), self.bn2, act())
        self.conv3 = nn.Sequential(nn.Conv2d(64 * 2, 128, kernel_size=1, bias=False), self.bn3, act())
        self.conv4 = nn.Sequential(nn.Conv2d(128 * 2, 256, kernel_size=1, bias=False), self.bn4, act())
        self.conv5 = nn.Sequential(nn.Conv1d(512, emb_dims, kernel_size=1, bias=False), self.bn5, act())

    def forward(self, xyz, backbone_list=None):
        """xyz (B,3,N) -> (xyz (B,3,N), per-point features (B,emb_dims,N))"""
        from pcr_amd import dgcnn_engine
        return dgcnn_engine.forward(self, xyz)
