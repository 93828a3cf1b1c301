"""Minimal ResNet-18 backbone pieces with torchvision-compatible parameter names.

The reference encoder subclasses ``torchvision.models.resnet.ResNet`` built from ``BasicBlock``s
(reference DDPM_DANRA_conditional/modules_DANRA_conditional.py:14,117,154) and its checkpoints carry
torchvision's parameter names (``conv1, bn1, layer1..4.{i}.{conv1,bn1,conv2,bn2,downsample.{0,1}}``).
This module provides just that: module creation order and names, kaiming-normal convolution
initialisation, and the block arithmetic for the autograd path.  No torchvision dependency.
"""
import torch.nn as nn

__all__ = ["BasicBlock", "ResNet", "norm_relu", "conv2d"]


def conv2d(conv, x):
    """``conv(x)``; on a CUDA tensor with gradients enabled the convolutions of the network (3x3, 8x8 stride 2, 1x1 stride
    2) run forward and backward in ``libdanra_hip.so`` where it has the shape (hip_engine.conv2d)."""
    if x.is_cuda:
        from . import hip_engine
        return hip_engine.conv2d(conv, x)
    return conv(x)


def norm_relu(bn, relu, x):
    """``relu(bn(x))`` (``relu`` None: just ``bn(x)``).  A train-mode BatchNorm2d on a CUDA fp32 tensor runs - with the
    ReLU fused in - forward and backward in ``libdanra_hip.so``; everything else is the two torch modules."""
    if x.is_cuda and type(bn) is nn.BatchNorm2d and bn.training:
        from . import hip_engine
        if hip_engine.batchnorm_train_supported(bn, x):
            return hip_engine.batchnorm_train(bn, x, relu is not None)
    y = bn(x)
    return y if relu is None else relu(y)


class BasicBlock(nn.Module):
    """conv3x3(stride) - BN - ReLU - conv3x3 - BN - (+ shortcut) - ReLU; ``expansion = 1``."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if self.downsample is None:
            shortcut = x
        elif isinstance(self.downsample, nn.Sequential) and len(self.downsample) == 2:
            shortcut = norm_relu(self.downsample[1], None, conv2d(self.downsample[0], x))
        else:
            shortcut = self.downsample(x)
        y = norm_relu(self.bn1, self.relu, conv2d(self.conv1, x))
        z = conv2d(self.conv2, y)
        if z.is_cuda and type(self.bn2) is nn.BatchNorm2d and self.bn2.training and shortcut.shape == z.shape:
            from . import hip_engine
            if hip_engine.batchnorm_train_supported(self.bn2, z) and hip_engine._native_dtype(shortcut):
                return hip_engine.batchnorm_add_relu_train(self.bn2, z, shortcut)  # bn2, + shortcut, ReLU: one kernel
        return self.relu(norm_relu(self.bn2, None, z) + shortcut)


class ResNet(nn.Module):
    """Creates ``conv1, bn1, relu, maxpool, layer1..layer4, avgpool, fc`` in torchvision's order (the
    encoder deletes the three members it does not use, as the reference does at modules:199)."""

    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _make_layer(self, block, planes, n_blocks, stride):
        out_planes = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != out_planes:
            shortcut = nn.Sequential(nn.Conv2d(self.inplanes, out_planes, 1, stride, bias=False),
                                     nn.BatchNorm2d(out_planes))
        blocks = [block(self.inplanes, planes, stride, shortcut)]
        self.inplanes = out_planes
        blocks += [block(self.inplanes, planes) for _ in range(1, n_blocks)]
        return nn.Sequential(*blocks)
