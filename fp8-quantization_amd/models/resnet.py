"""Full-precision ResNet-18 / ResNet-50 (BasicBlock / Bottleneck) with torchvision-compatible module names.

torchvision is not available in the build image, and the reference only needs torchvision for the
fp32 architecture (models/resnet_quantized.py:6-7).  State-dict keys match torchvision's
(`conv1`, `bn1`, `layer1.0.conv1`, ..., `layer2.0.downsample.0`, `fc`), so pretrained checkpoints load.
"""
import torch
from torch import nn


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, cout, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(cin, cout, stride)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(cout, cout)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + skip)


class Bottleneck(nn.Module):
    """1x1 reduce, 3x3 (carries the stride, as in torchvision's v1.5 layout), 1x1 expand by 4."""
    expansion = 4

    def __init__(self, cin, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _conv3x3(width, width, stride)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + skip)


class ResNet(nn.Module):
    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        widths, cin, stages = (64, 128, 256, 512), 64, []
        for i, (w, n) in enumerate(zip(widths, layers)):
            blocks = []
            for b in range(n):
                stride = 2 if (b == 0 and i > 0) else 1
                cout = w * block.expansion
                down = None
                if stride != 1 or cin != cout:
                    down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))
                blocks.append(block(cin, w, stride, down))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("no network in this environment: pass a checkpoint with --model-dir instead "
                           "of pretrained=True")
    return ResNet(BasicBlock, (2, 2, 2, 2), **kwargs)


def resnet50(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("no network in this environment: pass a checkpoint with --model-dir instead "
                           "of pretrained=True")
    return ResNet(Bottleneck, (3, 4, 6, 3), **kwargs)
