"""The reference's plugin seam: conf strings resolved by utils.general.get_class (utils/general.py:12-18) at
trainer/generate_ir_texture.py:55,64 and trainer/train_material.py:97,110,115.  Unmodified .conf files name the
reference's classes; the registry maps those dotted names onto the MI355X drop-ins.  Unknown names are imported
normally, so user classes keep working."""
import importlib

_REGISTRY = {
    "models.tracer_o3d_irt.TracerO3d": "texir_code_amd.models.TracerO3d",
    "models.mat_nvdiffrast.MaterialModel": "texir_code_amd.models.MaterialModel",
    "models.loss.RenderLoss": "texir_code_amd.loss.RenderLoss",
    "datasets.dataset.ImageCubeDerived": "texir_code_amd.datasets.ImageCubeDerived",
    "datasets.dataset.ImageCubeSyn": "texir_code_amd.datasets.ImageCubeSyn",
    # evaluation re-use: the tester runners' model and datasets
    "models.test_nvdiffrast.MaterialModel": "texir_code_amd.tester.test_model.MaterialModel",
    "datasets.dataset.ImageCubeNovel": "texir_code_amd.datasets.ImageCubeNovel",
    # NIrF slice
    "models.tracer_o3d_irrf.TracerO3d": "texir_code_amd.nirf.TracerO3dIrrF",
    "models.incidentNet.MatNetwork": "texir_code_amd.nirf.MatNetwork",
    "models.loss.IRFLoss": "texir_code_amd.nirf.IRFLoss",
    "datasets.dataset.MeshPoint": "texir_code_amd.datasets.MeshPoint",
    "datasets.dataset.ImageMeshPoint": "texir_code_amd.datasets.ImageMeshPoint",
}

# everything else the reference registers is a baseline / alternative lighting representation: out of scope
_OUT_OF_SCOPE = ("models.mat_nvdiffrast_", "models.mat_redner", "models.mat_mlp", "models.tracer_o3d.", "models.tracer_o3d_pil",
                 "models.incidentNet", "models.test_redner")


def get_class(kls):
    target = _REGISTRY.get(kls)
    if target is None:
        if any(kls.startswith(p) for p in _OUT_OF_SCOPE):
            raise NotImplementedError("%s is outside the IrT + material-estimation hot path this build covers" % kls)
        target = kls
    module, name = target.rsplit(".", 1)
    return getattr(importlib.import_module(module), name)
