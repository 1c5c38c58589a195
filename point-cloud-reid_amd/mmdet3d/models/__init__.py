from .builder import FUSIONMODELS, Registry, build_fusion_model, build_model
from .ReIDNet import ReIDNet, build_module, build_sequential, module_obj
from .backbone_net import Pointnet_Backbone
from .attention import corss_attention
from .lanegcn_nets import LinearRes
from .pointnet import PointNet

__all__ = ["FUSIONMODELS", "Registry", "build_model", "build_fusion_model", "ReIDNet", "module_obj",
           "build_module", "build_sequential", "Pointnet_Backbone", "corss_attention", "LinearRes", "PointNet"]
