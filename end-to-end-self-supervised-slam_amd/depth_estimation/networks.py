"""Drop-in for depth_estimation/networks.py: the monodepth2-style depth network of the reference with the
same class names, constructor arguments, forward signatures, output dictionaries and -- important for
checkpoints -- the same 150 state-dict keys (encoder.encoder.*, decoder.decoder.N.conv[.conv].*).

reference: depth_estimation/networks.py:16-57 (ResnetEncoder), :107-154 (DepthDecoder), :157-204 (ConvBlock /
Conv3x3 / Conv1x1), :206-221 (ScaleLayer, upsample), :224-292 (DispResNet_Indoor / Indoor_DepthDecoder).
torchvision is not required: the ResNet body is defined here (BasicBlock, [2,2,2,2] / [3,4,6,3]).

Every convolution goes through e2ehip.nn_ops.conv2d, which fuses eval-mode BatchNorm and the activation into
the convolution's epilogue and keeps activations in channels-last memory.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from e2ehip import conv as _e2e_conv
from e2ehip import nn_ops


def _bn_args(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


class _ConvBN(nn.Module):
    """Helper used by the ResNet body: conv (no bias) + eval-mode BatchNorm2d [+ residual] [+ ReLU] in ONE launch."""

    @staticmethod
    def run(conv, bn, x, relu, residual=None, in_norm=None):
        """relu?( BN(conv(x)) + residual? ).  A frozen BatchNorm (the refinement mode freezes every parameter whose name
        contains "bn", online_adaption.py:182-184) is folded into the convolution's epilogue as constants; one whose
        affine still trains (the `downsample.1` ones) runs e2ehip's affine kernel behind the convolution so that gamma / beta
        receive their gradients.

        Train-mode BatchNorm (batch statistics + running-average update): the reference only switches to eval() when
        MODEL.refinement_mode is set (online_adaption.py:175-184, train_depth.py:246-247); with the flag off the network stays in train
        mode.  That configuration is NOT the benchmarked path and has no native kernel: the convolution runs on the HIP kernel
        and the normalisation itself goes through the nn.BatchNorm2d module (running statistics, num_batches_tracked and the
        momentum=None cumulative average exactly as in the reference's modules) -- the one place where an ATen operator computes
        on behalf of the network ("train-mode BatchNorm")."""
        if bn.training:
            y = nn_ops.conv2d(x, conv.weight, None, conv.stride[0], conv.padding[0], "zeros", None, None, in_norm=in_norm)
            y = bn(y)       # the MODULE, not F.batch_norm: num_batches_tracked advances and momentum=None means the cumulative average
            if residual is not None:
                y = y + residual
            return torch.relu(y) if relu else y
        return nn_ops.conv2d(x, conv.weight, None, conv.stride[0], conv.padding[0], "zeros", "relu" if relu else None, _bn_args(bn),
                             residual=residual, in_norm=in_norm)


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
        idt = x if self.downsample is None else _ConvBN.run(self.downsample[0], self.downsample[1], x, False)
        out = _ConvBN.run(self.conv1, self.bn1, x, True)
        return _ConvBN.run(self.conv2, self.bn2, out, True, residual=idt)      # relu(bn2(conv2) + identity), one launch


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else _ConvBN.run(self.downsample[0], self.downsample[1], x, False)
        out = _ConvBN.run(self.conv1, self.bn1, x, True)
        out = _ConvBN.run(self.conv2, self.bn2, out, True)
        return _ConvBN.run(self.conv3, self.bn3, out, True, residual=idt)


class ResNet(nn.Module):
    """torchvision-compatible attribute / key layout (conv1, bn1, relu, maxpool, layer1-4, avgpool, fc)."""

    def __init__(self, block, layers, num_classes=1000, in_channels=3):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, 64, layers[0], 1)
        self.layer2 = self._stage(block, 128, layers[1], 2)
        self.layer3 = self._stage(block, 256, layers[2], 2)
        self.layer4 = self._stage(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)


_CFG = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
        101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


class ResNetMultiImageInput(ResNet):
    """ResNet whose first convolution takes num_input_images * 3 channels (reference: networks.py:60-83)."""

    def __init__(self, block, layers, num_classes=1000, num_input_images=1):
        super().__init__(block, layers, num_classes, in_channels=num_input_images * 3)


def _no_pretrained(pretrained):
    if pretrained:
        raise RuntimeError("ImageNet weights are downloaded by the reference (networks.py:100); there is no network here -- "
                           "load a checkpoint with load_state_dict instead (the key layout is identical)")


def resnet_multiimage_input(num_layers, pretrained=False, num_input_images=1):
    assert num_layers in [18, 50], "Can only run with 18 or 50 layer resnet"
    _no_pretrained(pretrained)
    block, layers = _CFG[num_layers]
    return ResNetMultiImageInput(block, layers, num_input_images=num_input_images)


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers, pretrained, num_input_images=1):
        super().__init__()
        if num_layers not in _CFG:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_input_images > 1:
            self.encoder = resnet_multiimage_input(num_layers, pretrained, num_input_images)
        else:
            _no_pretrained(pretrained)
            self.encoder = ResNet(*_CFG[num_layers])
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self._layout_group = _e2e_conv.group_layouts(self)      # GEMM copies of this model's weights are refreshed together (one launch)

    def forward(self, input_image):
        """(B,H,W,3) channels-last image in [0,1] -> the five feature maps (NCHW shape, NHWC memory)."""
        e = self.encoder
        # a permuted view of the NHWC frame IS a channels_last tensor; (x - 0.45) / 0.225 happens inside the stem's gather
        x = _ConvBN.run(e.conv1, e.bn1, input_image.permute(0, 3, 1, 2), True, in_norm=(0.45, 1.0 / 0.225))
        self.features = [x]
        x = nn_ops.max_pool_3x3_s2(x)
        for stage in (e.layer1, e.layer2, e.layer3, e.layer4):
            x = stage(x)
            self.features.append(x)
        return self.features


class Conv3x3(nn.Module):
    """reflection (or zero) pad 1 + 3x3 convolution (reference: networks.py:173-189)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad_mode = "reflect" if use_refl else "zeros"
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)     # kept for attribute compatibility
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x, act=None, skip=None, upsample=1):
        return nn_ops.conv2d(x, self.conv.weight, self.conv.bias, 1, 1, self.pad_mode, act, skip=skip, upsample=upsample)


class ConvBlock(nn.Module):
    """Conv3x3 + ELU, fused (reference: networks.py:157-170)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x, skip=None, upsample=1):
        return self.conv(x, act="elu", skip=skip, upsample=upsample)


class Conv1x1(nn.Module):
    """1x1 convolution used to learn a linear / affine depth scale (reference: networks.py:191-204)."""

    def __init__(self, in_channels, out_channels, init_value=0.5, bias=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=bias)
        self.conv.weight.data.fill_(init_value)

    def forward(self, x):
        return nn_ops.conv2d(x, self.conv.weight, self.conv.bias, 1, 0)


class ScaleLayer(nn.Module):
    def __init__(self, init_value=0.5):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([init_value]))

    def forward(self, x):
        return nn_ops.scale_layer(x, self.scale)


def upsample(x):
    """nearest-neighbour x2 (reference: networks.py:218-221)."""
    return nn_ops.upsample2_concat(x)


class _DecoderBase(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels, self.use_skips, self.upsample_mode, self.scales = num_output_channels, use_skips, "nearest", scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(cin, self.num_ch_dec[i])
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if (self.use_skips and i > 0) else 0)
            self.convs[("upconv", i, 1)] = ConvBlock(cin, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()
        self._layout_group = _e2e_conv.group_layouts(self)

    def _trunk(self, input_features):
        x = input_features[-1]
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x)
            # nearest x2 upsample + skip concatenation are gather arithmetic inside the next convolution
            x = self.convs[("upconv", i, 1)](x, skip=input_features[i - 1] if (self.use_skips and i > 0) else None, upsample=2)
            yield i, x


class DepthDecoder(_DecoderBase):
    """monodepth2 decoder: sigmoid disparities at every requested scale (reference: networks.py:107-154)."""

    def forward(self, input_features, index):
        self.outputs = {}
        for i, x in self._trunk(input_features):
            if i in self.scales:
                self.outputs[("disp", index, i)] = self.sigmoid(self.convs[("dispconv", i)](x))
        return self.outputs


class Indoor_DepthDecoder(_DecoderBase):
    """Indoor variant: only scale 0 is evaluated, disp = 10 * sigmoid(.) + 0.01 (reference: networks.py:241-292)."""

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__(num_ch_enc, scales, num_output_channels, use_skips)
        self.alpha, self.beta = 10, 0.01

    def forward(self, input_features, index):
        self.outputs = {}
        for i, x in self._trunk(input_features):
            if i in self.scales and i == 0:
                self.outputs[("disp", index, i)] = self.convs[("dispconv", i)](x, act="disp")   # alpha*sigmoid(.)+beta fused
        return self.outputs


class DispResNet_Indoor(nn.Module):
    def __init__(self, num_layers=18, pretrained=True):
        super().__init__()
        self.encoder = ResnetEncoder(num_layers=num_layers, pretrained=pretrained, num_input_images=1)
        self.decoder = Indoor_DepthDecoder(self.encoder.num_ch_enc)
        self._layout_group = _e2e_conv.group_layouts(self)      # encoder + decoder as ONE group (supersedes the two built above)

    def init_weights(self):
        pass

    def used_parameters(self):
        """The parameters that take part in forward(): everything except the classifier `encoder.encoder.fc` and the three
        dispconvs of the unused scales (networks.py:271-272 allocates four, :289-290 evaluates scale 0 only) -- these never
        receive a gradient."""
        skip = {id(p) for p in self.encoder.encoder.fc.parameters()}
        for s in self.decoder.scales:
            if s != 0:
                skip |= {id(p) for p in self.decoder.convs[("dispconv", s)].parameters()}
        return [p for p in self.parameters() if id(p) not in skip]

    def forward(self, x, index):
        return self.decoder(self.encoder(x), index)
