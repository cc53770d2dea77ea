"""Full-precision MobileNetV2 (width 1.0) whose module tree matches the checkpoint layout the
reference loads (`features.N.conv.K.*`, `classifier.1.*`; /root/reference/models/mobilenet_v2.py).

Architecture (Sandler et al. 2018): 3x3 stem /2, seven inverted-residual stages given as
(expansion t, channels c, repeats n, stride s), 1x1 head to 1280, global pool, dropout, linear.
"""
import math

from torch import nn

__all__ = ["MobileNetV2", "InvertedResidual"]

_STAGES = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2),
           (6, 320, 1, 1))


def _cba(cin, cout, k, stride, groups=1):
    """conv -> batch norm -> ReLU6, as three consecutive entries (so BN folding sees them)."""
    return [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout),
            nn.ReLU6(inplace=True)]


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        assert stride in (1, 2)
        self.stride = stride
        hidden = round(inp * expand_ratio)
        self.use_res_connect = stride == 1 and inp == oup
        layers = [] if expand_ratio == 1 else _cba(inp, hidden, 1, 1)      # pointwise expansion
        layers += _cba(hidden, hidden, 3, stride, groups=hidden)            # depthwise
        layers += [nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]   # linear projection
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNetV2(nn.Module):
    def __init__(self, n_class=1000, input_size=224, width_mult=1.0, dropout=0.0):
        super().__init__()
        assert input_size % 32 == 0
        cin = int(32 * width_mult)
        self.last_channel = int(1280 * width_mult) if width_mult > 1.0 else 1280
        feats = [nn.Sequential(*_cba(3, cin, 3, 2))]
        for t, c, n, s in _STAGES:
            cout = int(c * width_mult)
            for i in range(n):
                feats.append(InvertedResidual(cin, cout, s if i == 0 else 1, expand_ratio=t))
                cin = cout
        feats.append(nn.Sequential(*_cba(cin, self.last_channel, 1, 1)))
        feats.append(nn.AvgPool2d(input_size // 32))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(dropout), nn.Linear(self.last_channel, n_class))
        self._init()

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.flatten(1))

    def _init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()
