"""qbm_image_classification_amd -- MI355X-native Gibbs / contrastive-divergence path of
Mark-Seebode/QBM-Image-Classification (ClassificationRBM + Disc_QBM sampler), hand-written HIP
behind a C ABI (include/qbm_hip.h).  See DESIGN.md."""
from ._lib import QbmHipError  # noqa: F401

__all__ = ["QbmHipError", "ClassificationRBM", "HipRbmOps"]


def __getattr__(name):  # lazy: importing the package must not require torch / the GPU
    if name == "ClassificationRBM":
        from .classification_rbm import ClassificationRBM

        return ClassificationRBM
    if name == "HipRbmOps":
        from .rbm_ops import HipRbmOps

        return HipRbmOps
    raise AttributeError(name)
