"""CPU restatement (torch, fp32) of the monocular depth CNN on the path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

* Decoder, ConvBlock/Conv3x3 (reflection pad), nearest x2 upsample, skip wiring, the
  `10*sigmoid+0.01` head and the NHWC->NCHW + (x-0.45)/0.225 stem follow the reference's
  depth_estimation/networks.py and are PINNED by tests/golden/g7 (reference DispResNet_Indoor
  executed in the build container).
* The ResNet-18 body restates torchvision.models.resnet18 (BasicBlock, [2,2,2,2]); torchvision
  is not vendored in /root/reference and not installed -> that part is PARITY UNPINNED
  (architecture is the published one; key names checked against networks.py's usage).

The functional form works on a flat state dict with the reference's 150 keys
(`encoder.encoder.*`, `decoder.decoder.N.conv[.conv].{weight,bias}`).
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

NUM_CH_ENC = (64, 64, 128, 256, 512)      # networks.py:23
NUM_CH_DEC = (16, 32, 64, 128, 256)       # networks.py:253


# ---- torchvision-style modules: used (a) as the stub the reference's networks.py imports in
# ---- make_golden.py and (b) to create a correctly-named random state dict. ------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):           # only so `models.resnet.Bottleneck` resolves (networks.py:96)
    expansion = 4


class ResNet(nn.Module):
    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def resnet18(pretrained=False, **kw):
    if pretrained:
        raise RuntimeError("no network: pretrained weights are unavailable")
    return ResNet(BasicBlock, (2, 2, 2, 2))


def decoder_layout():
    """[(key_prefix, cin, cout)] in ModuleList order: upconv(4,0),(4,1),(3,0)...(0,1), dispconv 0..3.
    reference: networks.py:258-274."""
    out, idx = [], 0
    for i in range(4, -1, -1):
        cin = NUM_CH_ENC[-1] if i == 4 else NUM_CH_DEC[i + 1]
        out.append((f"decoder.decoder.{idx}.conv.conv", cin, NUM_CH_DEC[i])); idx += 1
        cin = NUM_CH_DEC[i] + (NUM_CH_ENC[i - 1] if i > 0 else 0)
        out.append((f"decoder.decoder.{idx}.conv.conv", cin, NUM_CH_DEC[i])); idx += 1
    for s in range(4):
        out.append((f"decoder.decoder.{idx}.conv", NUM_CH_DEC[s], 1)); idx += 1
    return out


def random_state_dict(seed=0, bn_noise=True):
    """A random 150-key state dict with non-trivial BN statistics (so eval-BN is exercised)."""
    g = torch.Generator().manual_seed(seed)
    enc = resnet18()
    sd = OrderedDict()
    for k, v in enc.state_dict().items():
        v = v.clone()
        if v.dtype.is_floating_point:
            if k.endswith("running_var"):
                v = 0.5 + torch.rand(v.shape, generator=g)
            elif k.endswith("running_mean"):
                v = 0.1 * torch.randn(v.shape, generator=g)
            elif ("bn" in k or "downsample.1" in k) and bn_noise:
                v = (1.0 if k.endswith("weight") else 0.0) + 0.1 * torch.randn(v.shape, generator=g)
            elif k.startswith("fc"):
                v = 0.01 * torch.randn(v.shape, generator=g)
            else:
                fan_out = v.shape[0] * v.shape[2] * v.shape[3]
                v = torch.randn(v.shape, generator=g) * math.sqrt(2.0 / fan_out)
        sd["encoder.encoder." + k] = v
    for prefix, cin, cout in decoder_layout():
        bound = 1.0 / math.sqrt(cin * 9)
        sd[prefix + ".weight"] = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) * bound
        sd[prefix + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound
    return sd


# ---- functional forward ---------------------------------------------------------------------
BN_TRAINING = [False]     # True: batch statistics + running-average update (nn.BatchNorm2d in train mode, momentum 0.1) -- the
                          # configuration MODEL.refinement_mode = False leaves the reference in (train_depth.py:246-247)


def _bn_eval(x, sd, p):
    if BN_TRAINING[0]:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _block(x, sd, p, stride, has_down):
    idt = x
    if has_down:
        idt = _bn_eval(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
    out = F.relu(_bn_eval(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
    out = _bn_eval(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    return F.relu(out + idt)


def encoder_forward(sd, image_nhwc):
    """(B,H,W,3) -> 5 feature maps.  reference: networks.py:44-57 (BN always in eval mode on the
    path: online_adaption.py:175-184)."""
    e = "encoder.encoder."
    x = (image_nhwc.permute(0, 3, 1, 2) - 0.45) / 0.225
    x = F.relu(_bn_eval(F.conv2d(x, sd[e + "conv1.weight"], None, 2, 3), sd, e + "bn1"))
    feats = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = _block(x, sd, f"{e}layer{li}.0", stride, li > 1)
        x = _block(x, sd, f"{e}layer{li}.1", 1, False)
        feats.append(x)
    return feats


def _conv3x3_reflect(x, w, b):
    """reference: networks.py:173-189 (ReflectionPad2d(1) + Conv2d(3))."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def decoder_forward(sd, feats):
    """reference: networks.py:277-292 (only scale 0 is evaluated; disp = 10*sigmoid(.)+0.01)."""
    lay = decoder_layout()
    x = feats[-1]
    idx = 0
    for i in range(4, -1, -1):
        p = lay[idx][0]; idx += 1
        x = F.elu(_conv3x3_reflect(x, sd[p + ".weight"], sd[p + ".bias"]))
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if i > 0:
            x = torch.cat([x, feats[i - 1]], 1)
        p = lay[idx][0]; idx += 1
        x = F.elu(_conv3x3_reflect(x, sd[p + ".weight"], sd[p + ".bias"]))
    p = lay[10][0]
    return 10 * torch.sigmoid(_conv3x3_reflect(x, sd[p + ".weight"], sd[p + ".bias"])) + 0.01


def disp_forward(sd, image_nhwc):
    """DispResNet_Indoor.forward.  reference: networks.py:234-238."""
    return decoder_forward(sd, encoder_forward(sd, image_nhwc))


def trainable_keys(sd):
    """Keys that receive a gradient on the refinement path: everything that is a parameter,
    minus names containing "bn" (online_adaption.py:182-184), minus resnet fc and the unused
    dispconv1-3 (grad None: networks.py:289-290)."""
    keys = []
    for k in sd:
        if "running_" in k or "num_batches" in k or "bn" in k:
            continue
        if k.startswith("encoder.encoder.fc") or any(k.startswith(f"decoder.decoder.{j}.") for j in (11, 12, 13)):
            continue
        keys.append(k)
    return keys
