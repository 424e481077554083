"""ImageNet-style ResNets with the distillation feature contract (reference: models/resnet_imagenet.py:227-250).

`model(x, is_feat=True) -> ([stem, layer1, layer2, layer3, layer4, pooled], logits)`; the MoMA loop uses the last
entry (pooled [B, 512 * expansion]).  Parameter names follow the torchvision layout (conv1, bn1, layerN.M.convK,
downsample.0/1, fc), so a locally saved checkpoint of either the reference or torchvision loads with
`--std_pre / --tec_pre <path>`.  Convolutions and BatchNorm stay on MIOpen (backbones out of scope)."""
import torch
import torch.nn as nn


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, width, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv3x3(cin, width, stride), nn.BatchNorm2d(width)
        self.conv2, self.bn2 = _conv3x3(width, width), nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, width, 1, bias=False), nn.BatchNorm2d(width)
        self.conv2, self.bn2 = _conv3x3(width, width, stride), nn.BatchNorm2d(width)
        self.conv3, self.bn3 = nn.Conv2d(width, width * 4, 1, bias=False), nn.BatchNorm2d(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, block, depths, num_classes=1000):
        super().__init__()
        self.cin = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, 64, depths[0], 1)
        self.layer2 = self._stage(block, 128, depths[1], 2)
        self.layer3 = self._stage(block, 256, depths[2], 2)
        self.layer4 = self._stage(block, 512, depths[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _stage(self, block, width, n, stride):
        down = None
        if stride != 1 or self.cin != width * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.cin, width * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(width * block.expansion))
        blocks = [block(self.cin, width, stride, down)]
        self.cin = width * block.expansion
        blocks += [block(self.cin, width) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def get_feat_modules(self):
        return nn.ModuleList([self.conv1, self.bn1, self.relu, self.maxpool, self.layer1, self.layer2, self.layer3,
                              self.layer4, self.fc])

    def forward(self, x, is_feat=False):
        feats = []
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        feats.append(x)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = stage(x)
            feats.append(x)
        x = torch.flatten(self.avgpool(x), 1)
        feats.append(x)
        logits = self.fc(x)
        return (feats, logits) if is_feat else logits


def ResNet18(num_classes=1000, **_):
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes)


def ResNet34(num_classes=1000, **_):
    return ResNet(BasicBlock, [3, 4, 6, 3], num_classes)


def ResNet50(num_classes=1000, **_):
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes)


def resnet101(num_classes=1000, **_):
    return ResNet(Bottleneck, [3, 4, 23, 3], num_classes)
