"""CIFAR-style ResNet (6n+2 / 9n+2 layers) with the feature-list contract the MoMA loop needs:
`model(x, is_feat=True) -> ([f0, f1, f2, f3, pooled], logits)` and `get_feat_modules()`.

Own implementation; parameter names (conv1, bn1, layer{1,2,3}.N.{conv1,bn1,conv2,bn2,downsample.0/1},
fc) follow the reference's checkpoints (models/resnet.py:118-186) so its state_dicts load unchanged.
Backbones stay on PyTorch-ROCm / MIOpen (out of scope as kernels).
"""
import torch.nn as nn


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 3, stride), nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _conv(planes, planes, 3), nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += skip
        return self.relu(y)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y += skip
        return self.relu(y)


class ResNet(nn.Module):
    def __init__(self, depth, num_filters, block_name="BasicBlock", num_classes=10):
        super().__init__()
        if block_name.lower() == "basicblock":
            assert (depth - 2) % 6 == 0, "basicblock depth must be 6n+2"
            n, block = (depth - 2) // 6, BasicBlock
        elif block_name.lower() == "bottleneck":
            assert (depth - 2) % 9 == 0, "bottleneck depth must be 9n+2"
            n, block = (depth - 2) // 9, Bottleneck
        else:
            raise ValueError("block_name should be Basicblock or Bottleneck")
        self.inplanes = num_filters[0]
        self.conv1 = _conv(3, num_filters[0], 3)
        self.bn1 = nn.BatchNorm2d(num_filters[0])
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._stage(block, num_filters[1], n, 1)
        self.layer2 = self._stage(block, num_filters[2], n, 2)
        self.layer3 = self._stage(block, num_filters[3], n, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(num_filters[3] * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, block, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, stride),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def get_feat_modules(self):
        return nn.ModuleList([self.conv1, self.bn1, self.relu, self.layer1, self.layer2, self.layer3, self.fc])

    def forward(self, x, is_feat=False):
        f0 = self.relu(self.bn1(self.conv1(x)))
        f1 = self.layer1(f0)
        f2 = self.layer2(f1)
        f3 = self.layer3(f2)
        f4 = self.avgpool(f3).view(x.size(0), -1)
        out = self.fc(f4)
        return ([f0, f1, f2, f3, f4], out) if is_feat else out


def _mk(depth, filters):
    def ctor(**kw):
        return ResNet(depth, filters, "basicblock", **kw)
    return ctor


resnet8 = _mk(8, [16, 16, 32, 64])
resnet14 = _mk(14, [16, 16, 32, 64])
resnet20 = _mk(20, [16, 16, 32, 64])
resnet32 = _mk(32, [16, 16, 32, 64])
resnet44 = _mk(44, [16, 16, 32, 64])
resnet56 = _mk(56, [16, 16, 32, 64])
resnet110 = _mk(110, [16, 16, 32, 64])
resnet8x4 = _mk(8, [32, 64, 128, 256])
resnet32x4 = _mk(32, [32, 64, 128, 256])
