"""Host-side mirror of the reference's model factory (models/__init__.py:16-132).

Same public names (``activate``, ``ModelBuilder.build_sound / build_frame / build_synthesizer /
build_criterion / weights_init``), same argument meaning, same exceptions, so the reference's
``main.py:606-629`` can build its nets from this package unchanged.  What is built is different:
the U-Net, fusion and losses are launch sequences over libavsep_gfx950.so.
``build_motion`` (mmaction SlowFast + a private checkpoint, :94-118) is out of scope.
"""
import torch

from .synthesizer_net import InnerProd, Bias
from .audio_net import Unet
from .vision_net import ResnetFC, ResnetDilated
from .criterion import BCELoss, L1Loss, L2Loss, PitWrapper
from .fusion_net import get_fusion_net
from .attention_net import get_attmodule, AttModel, MatchAtt

_ACTIVATIONS = {
    "sigmoid": torch.sigmoid,
    "softmax": lambda x: torch.softmax(x, dim=1),
    "relu": torch.relu,
    "tanh": torch.tanh,
    "no": lambda x: x,
}
_UNET_DOWNS = {"unet5": 5, "unet6": 6, "unet7": 7}
_FRAME_NETS = {"resnet18fc": ResnetFC, "resnet18dilated": ResnetDilated}
_CRITERIA = {"bce": BCELoss, "l1": L1Loss, "l2": L2Loss}


def activate(x, activation):
    fn = _ACTIVATIONS.get(activation)
    if fn is None:
        raise Exception("Unkown activation!")  # (sic) same message as the reference
    return fn(x)


def _maybe_load(net, weights, what):
    if len(weights) > 0:
        print(f"Loading weights for {what}")
        net.load_state_dict(torch.load(weights))
    return net


class ModelBuilder:
    def weights_init(self, m):
        """Conv ~ N(0, 1e-3), BatchNorm weight ~ N(1, 0.02) / bias 0, Linear ~ N(0, 1e-4); matched by
        class-name substring like the reference, so the HIP parameter holders are covered."""
        name = type(m).__name__
        if "Conv" in name:
            m.weight.data.normal_(0.0, 0.001)
        elif "BatchNorm" in name:
            m.weight.data.normal_(1.0, 0.02)
            m.bias.data.fill_(0)
        elif "Linear" in name:
            m.weight.data.normal_(0.0, 0.0001)

    def build_sound(self, arch="unet5", fc_dim=64, weights="", fusion_type="con_motion", att_type="cos",
                    extra_size=None):
        # extra_size: the SoP++ driver's extra keyword (SoP++/main.py:727-732) selects the basis U-Net
        if arch not in _UNET_DOWNS:
            raise Exception("Architecture undefined!")
        net = Unet(fc_dim=fc_dim, num_downs=_UNET_DOWNS[arch], fusion_type=fusion_type, att_type=att_type,
                   extra_size=extra_size)
        net.apply(self.weights_init)
        return _maybe_load(net, weights, "net_sound")

    def build_frame(self, arch="resnet18", fc_dim=64, pool_type="avgpool", weights=""):
        # The reference hard-codes pretrained=True (an ImageNet download).  Offline the trunk keeps
        # PyTorch's default init unless `weights` names a checkpoint; weights_init is not applied
        # to net_frame in the reference either.
        if arch not in _FRAME_NETS:
            raise Exception("Architecture undefined!")
        net = _FRAME_NETS[arch](None, fc_dim=fc_dim, pool_type=pool_type)
        return _maybe_load(net, weights, "net_frame")

    def build_synthesizer(self, arch, fc_dim=64, weights=""):
        if arch == "linear":
            net = InnerProd(fc_dim=fc_dim)
        elif arch == "bias":
            net = Bias()
        else:
            raise Exception("Architecture undefined!")
        net.apply(self.weights_init)
        return _maybe_load(net, weights, "net_synthesizer")

    def build_motion(self):
        raise NotImplementedError("build_motion needs mmaction and a private checkpoint: out of scope")

    def build_criterion(self, arch, use_pit=False):
        if arch not in _CRITERIA:
            raise Exception("Architecture undefined!")
        if use_pit:  # the reference ignores `arch` here and always wraps BCE (:130-131)
            return PitWrapper(torch.nn.functional.binary_cross_entropy)
        return _CRITERIA[arch]()
