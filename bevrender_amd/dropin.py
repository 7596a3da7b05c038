"""Make a checkout of the reference import the MI355X path, without editing its files.

    import bevrender_amd.dropin; bevrender_amd.dropin.install()      # before the reference's own imports

After this `from model.bevrender import BEVRender`, `from loss.contrastive_loss import ContrastiveLoss`,
`from loss.lift_loss import LiftedStructureLoss` and `from loss.triplet_loss_metric import TripletLossMetricLearning`
(reference train.py:20-23) resolve to this package.  Only what this package mirrors is aliased: the `model` package
is mirrored file by file, so it is replaced as a whole; of `loss` only the three retrieval-loss submodules are
aliased and the reference's own `loss` package stays in place, so `loss.mse_loss`, `loss.l1_loss` and
`loss.cross_entropy_loss` (train.py:17-19) keep importing from the reference.
"""
import importlib
import sys

MODEL_MODULES = ("bevrender", "encoder", "SCA", "SCA_deform_attn", "TSA", "TSA_deform_attn", "bev_cmr_proj",
                 "img_backbone", "decoder_img_render", "model_utils", "feedforward")
LOSS_MODULES = ("contrastive_loss", "lift_loss", "triplet_loss_metric")


def install() -> None:
    pkg = importlib.import_module("bevrender_amd.model")
    sys.modules["model"] = pkg
    for name in MODEL_MODULES:
        sys.modules[f"model.{name}"] = importlib.import_module(f"bevrender_amd.model.{name}")
    for name in LOSS_MODULES:
        sys.modules[f"loss.{name}"] = importlib.import_module(f"bevrender_amd.loss.{name}")
