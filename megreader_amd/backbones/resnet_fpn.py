"""Mirror of reference backbones/resnet_fpn.py:6-38 (including quirk Q14: Resnet34FPN builds a ResNet-50).
The reference defaults resnet_pretrained=True and downloads ImageNet weights at construction; without a network
that raises here -- pass backbone_args: {resnet_pretrained: false} in an additive YAML."""
from .resnet import resnet18, resnet50, resnet101, resnet152
from .fpn_top_down import FPNTopDown
from .feature_pyramid import FeaturePyramid


def Resnet18FPN(resnet_pretrained=True):
    return FeaturePyramid(resnet18(pretrained=resnet_pretrained), FPNTopDown([512, 256, 128, 64], 256))


def Resnet34FPN(resnet_pretrained=True):
    return FeaturePyramid(resnet50(pretrained=resnet_pretrained), FPNTopDown([2048, 1024, 512, 256], 256))


def Resnet50FPN(resnet_pretrained=True):
    return FeaturePyramid(resnet50(pretrained=resnet_pretrained), FPNTopDown([2048, 1024, 512, 256], 256))


def Resnet101FPN(resnet_pretrained=True):
    return FeaturePyramid(resnet101(pretrained=resnet_pretrained), FPNTopDown([2048, 1024, 512, 256], 256))


def Resnet152FPN(resnet_pretrained=True):
    return FeaturePyramid(resnet152(pretrained=resnet_pretrained), FPNTopDown([2048, 1024, 512, 256], 256))
