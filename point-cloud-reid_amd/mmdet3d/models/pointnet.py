"""PointNet encoder with input (STN3d) and feature (STNkd) transforms returning per-point
1024-d features without global pooling (reference: mmdet3d/models/pointnet.py -- STN3d :10-45,
STNkd :48-85, PointNetEncoder :88-127, PointNet :139-149).  Parameter names match the reference."""
import torch
import torch.nn as nn


class _STN(nn.Module):
    def __init__(self, cin, k):
        super().__init__()
        self.conv1 = nn.Conv1d(cin, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k * k)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)
        self.k = k


class STN3d(_STN):
    def __init__(self, channel):
        super().__init__(channel, 3)


class STNkd(_STN):
    def __init__(self, k=64):
        super().__init__(k, k)


class PointNetEncoder(nn.Module):
    def __init__(self, global_feat=True, feature_transform=False, channel=3):
        super().__init__()
        if channel != 3:
            raise NotImplementedError("the ReID configs build PointNet with normal_channel=False (xyz only)")
        self.stn = STN3d(channel)
        self.conv1 = nn.Conv1d(channel, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.global_feat = global_feat
        self.feature_transform = feature_transform
        if self.feature_transform:
            self.fstn = STNkd(k=64)

    def forward(self, xyz):
        from pcr_amd import pointnet_engine
        return pointnet_engine.encoder_forward(self, xyz)


class PointNet(nn.Module):
    def __init__(self, k=40, normal_channel=True):
        super().__init__()
        self.feat = PointNetEncoder(global_feat=True, feature_transform=True, channel=6 if normal_channel else 3)

    def forward(self, x, backbone_list):
        """x (B,3,N) -> (xyz (B,3,N), per-point features (B,1024,N))"""
        return self.feat(x)
