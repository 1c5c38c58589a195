"""Drop-in mirror of the slice of the reference's `mmdet3d` package that the siamese point-cloud
ReID hot path uses (bentherien/point-cloud-reid): `mmdet3d.models` (FUSIONMODELS, build_model,
ReIDNet, module_obj/build_module and the point backbones) and `mmdet3d.ops` (the point ops).
Everything is backed by libpcr_hip.so (hand-written gfx950 kernels); nothing here needs mmcv,
mmdet or CUDA.  Components of the reference outside SURVEY.md section 8 are intentionally absent."""
__version__ = "0.1.0+pcr.gfx950"
