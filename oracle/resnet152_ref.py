"""ResNet-152 trunk as plain torch.nn -- oracle for capnet_trunk_forward. TEST INFRASTRUCTURE.

Restates torchvision==0.2.2.post3 `models.resnet152` (pinned in stylenet/requirements.txt:4),
the third-party network EncoderCNN wraps at stylenet/model.py:15-18 (children()[:-1]) and
stylenet/model_att.py:15-18 (children()[:-2]). torchvision is absent from this image, so this
follows its published definition: conv 7x7/2 p3 (no bias) -> BN -> ReLU -> MaxPool 3x3/2 p1 ->
Bottleneck x [3, 8, 36, 3] (planes 64/128/256/512, expansion 4, 1x1 -> 3x3 carrying the stride
-> 1x1, downsample = 1x1 conv(stride) + BN on the first block of each layer) -> AvgPool(7).
Init: conv kaiming_normal_(fan_out, relu), BN weight 1 / bias 0. PARITY UNPINNED vs the
reference (no fixture exists there); module/key names match torchvision's.
"""
import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


def resnet152_children(with_avgpool=True):
    """nn.Sequential equal to `nn.Sequential(*list(resnet152().children())[:-1 or -2])`."""
    mods = [nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
            nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=1)]
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 8, 36, 3))):
        layer = []
        for b in range(blocks):
            stride = 2 if (b == 0 and li > 0) else 1
            ds = None
            if b == 0:
                ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes * 4))
            layer.append(Bottleneck(inplanes, planes, stride, ds))
            inplanes = planes * 4
        mods.append(nn.Sequential(*layer))
    if with_avgpool:
        mods.append(nn.AvgPool2d(7, stride=1))
    seq = nn.Sequential(*mods)
    for m in seq.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return seq


class EncoderCNNRef(nn.Module):
    """EncoderCNN.forward of stylenet/model.py:22-27 over the restated trunk."""

    def __init__(self, embed_size):
        super().__init__()
        self.resnet = resnet152_children(True)
        self.linear = nn.Linear(2048, embed_size)
        self.bn = nn.BatchNorm1d(embed_size, momentum=0.01)

    def forward(self, images):
        with torch.no_grad():
            features = self.resnet(images)
        features = features.reshape(features.size(0), -1)
        return self.bn(self.linear(features))


class EncoderCNNAttRef(nn.Module):
    """EncoderCNN.forward of stylenet/model_att.py:22-29."""

    def __init__(self, encoded_image_size=14):
        super().__init__()
        self.resnet = resnet152_children(False)
        self.adaptive_pool = nn.AdaptiveAvgPool2d((encoded_image_size, encoded_image_size))

    def forward(self, images):
        with torch.no_grad():
            features = self.resnet(images)
        features = self.adaptive_pool(features)
        return features.permute(0, 2, 3, 1)
