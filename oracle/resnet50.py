"""Oracle: ResNet-50 backbone + whole train step on PyTorch CPU (ATen fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference builds its backbone with torchvision (utils/backbones.py:16-18:
`resnet50(weights=...)`, `fc = nn.Linear(2048, FEATURE_DIM)`).  torchvision is a
third-party dependency that is NOT vendored in /root/reference, is unpinned
(requirement.txt lists neither torch nor torchvision) and is absent from this image,
and its pretrained weights are a network download.  So this file restates
torchvision's published ResNet-50 **v1.5** topology (stride on the 3x3 conv of each
down-sampling Bottleneck) from torch.nn primitives, with torchvision's state-dict
key names.  PARITY UNPINNED at this boundary: no reference fixture covers it.

The rest of the step follows the reference:
  model wrapper   utils/criterion.py:303-325 (ArcFaceNet & siblings)
  train step      utils/model_utils.py:176-187 (fwd, CE, zero_grad, backward, SGD step)
  SGD             utils/model_utils.py:557  (momentum 0.9, wd 5e-4 on all params)
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
FEATURE_DIM = 512


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class ResNet50(nn.Module):
    def __init__(self, feature_dim=FEATURE_DIM):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make(64, LAYERS[0], 1)
        self.layer2 = self._make(128, LAYERS[1], 2)
        self.layer3 = self._make(256, LAYERS[2], 2)
        self.layer4 = self._make(512, LAYERS[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, feature_dim)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class TorchHead(nn.Module):
    """Autograd restatement of the heads, used only to drive a whole-step oracle
    (the closed-form numpy version in oracle/heads.py is the pinned one; this class is
    checked against it in tests/test_oracle_heads.py).  After a forward, `loss_g` holds MagFace's
    magnitude loss (0 for the other kinds) and `row_margin` the elastic heads' margins; setting
    `next_margin` [N] before a forward replaces the torch.normal draw (criterion.py:1002, 1113)."""

    def __init__(self, kind, feat_dim, num_classes, hyper):
        super().__init__()
        from . import heads as H
        self.kind, self.hyper, self.H = kind, hyper, H
        self.state = H.HeadState()
        shape = (num_classes, feat_dim) if H.weight_is_cd(kind) else (feat_dim, num_classes)
        self.weight = nn.Parameter(torch.empty(shape))
        self.loss_g, self.row_margin, self.next_margin, self.elastic_std = 0, None, None, 0.0125
        if kind in (H.ARC, H.SPHERE, H.VPL):
            nn.init.xavier_uniform_(self.weight)                       # criterion.py:244,37,657
        elif kind in (H.COS, H.MV_AM, H.MV_ARC, H.ADA, H.MAG):
            self.weight.data.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)  # :152,367,833,1216
        else:
            nn.init.normal_(self.weight, std=0.01)                     # :514,973,1080

    def forward(self, x, labels):
        import math
        H, hy = self.H, self.hyper
        rows = torch.arange(x.shape[0])
        if H.weight_is_cd(self.kind):
            c = F.linear(F.normalize(x), F.normalize(self.weight))
        else:
            c = torch.mm(F.normalize(x, dim=1), F.normalize(self.weight, dim=0))
        onehot = torch.zeros_like(c).scatter_(1, labels.view(-1, 1), 1)
        if self.kind == H.ARC:
            sine = torch.sqrt((1.0 - c * c).clamp(1e-9, 1.0))
            phi = c * math.cos(hy.m) - sine * math.sin(hy.m)
            phi = torch.where(c > math.cos(math.pi - hy.m), phi, c - math.sin(math.pi - hy.m) * hy.m)
            return c * hy.s, (onehot * phi + (1 - onehot) * c) * hy.s
        if self.kind == H.COS:
            c = c.clamp(-1 + 1e-4, 1 - 1e-4)
            return c * hy.s, (c - onehot * hy.m) * hy.s
        if self.kind == H.SPHERE:
            st = self.state
            st.iter += 1
            st.lamb = max(hy.lambda_min, hy.base * (1 + hy.gamma * st.iter) ** (-hy.power))
            c = c.clamp(-1, 1)
            k = (2 * c.detach().acos() / math.pi).floor()
            phi = ((-1.0) ** k) * (2 * c * c - 1) - 2 * k
            nrm = torch.norm(x, p=2, dim=1, keepdim=True)
            return c * nrm, (onehot * (phi - c) / (1 + st.lamb) + c) * nrm
        if self.kind in (H.MV_AM, H.MV_ARC):                           # criterion.py:414-441
            c = c.clamp(-1 + 1e-7, 1 - 1e-7)
            ty = c[rows, labels].view(-1, 1)
            if self.kind == H.MV_AM:
                fin, thr = torch.where(ty > hy.m, ty - hy.m, ty), ty - hy.m
            else:
                thr = ty * math.cos(hy.m) - torch.sqrt(1.0 - ty ** 2 + 1e-9) * math.sin(hy.m)
                fin = torch.where(ty > 0.0, thr, ty)
            z = torch.where(c > thr, hy.mv_weight * c + (hy.mv_weight - 1.0), c)
            return c * hy.s, z.scatter(1, labels.view(-1, 1), fin) * hy.s
        if self.kind == H.ADA:                                         # criterion.py:866-899
            st, eps = self.state, 1e-3
            c = c.clamp(-1 + eps, 1 - eps)
            with torch.no_grad():
                safe = torch.norm(x, p=2, dim=1, keepdim=True).clamp(0.001, 100)
                st.batch_mean = float(safe.mean() * hy.t_alpha + (1 - hy.t_alpha) * st.batch_mean)
                st.batch_std = float(safe.std() * hy.t_alpha + (1 - hy.t_alpha) * st.batch_std)
                ms = ((safe - st.batch_mean) / (st.batch_std + eps) * hy.h).clamp(-1, 1)
            theta_m = (c.acos() + onehot * (hy.m * ms * -1)).clamp(eps, math.pi - eps)
            return c * hy.s, (theta_m.cos() - onehot * (hy.m + hy.m * ms)) * hy.s
        if self.kind in (H.ELASTIC_ARC, H.ELASTIC_COS):                # criterion.py:997-1015, 1108-1135
            c = c.clamp(-1 + 1e-7, 1 - 1e-7)
            if self.next_margin is not None:
                margin, self.next_margin = self.next_margin.view(-1, 1).to(c.dtype), None
            else:
                margin = torch.normal(mean=hy.m, std=self.elastic_std, size=(x.shape[0], 1))
                margin = margin.clamp(hy.m - self.elastic_std, hy.m + self.elastic_std)
            self.row_margin = margin.view(-1)
            if self.kind == H.ELASTIC_COS:
                return c * hy.s, (c - onehot * margin) * hy.s
            fin = (c[rows, labels].acos().view(-1, 1) + margin).clamp(0, math.pi).cos()
            return c * hy.s, c.scatter(1, labels.view(-1, 1), fin) * hy.s
        if self.kind == H.VPL:                                         # criterion.py:699-744
            st = self.state
            if hy.memory_on:
                with torch.no_grad():
                    if st.mem is None:
                        st.mem, st.life = torch.zeros_like(self.weight), torch.zeros(self.weight.shape[0])
                    for cls in torch.unique(labels):
                        st.mem[cls] = x.detach()[labels == cls].mean(dim=0)
                        st.life[cls] = hy.delta
                    st.life = st.life - 1
                    act = (st.life > 0).float().unsqueeze(0)
                c_mem = F.linear(F.normalize(x), F.normalize(st.mem, dim=1))
                cos1 = (1 - act * hy.lamda) * c + act * hy.lamda * c_mem
                cos2 = (1 - act * hy.lamda) * c + act * hy.lamda * 1.0
                c = onehot * cos2 + (1.0 - onehot) * cos1
            c = c.clamp(-1 + 1e-7, 1 - 1e-7)
            phi = c * math.cos(hy.m) - torch.sqrt(1.0 - c ** 2 + 1e-9) * math.sin(hy.m)
            if hy.easy_margin:
                phi = torch.where(c > 0, phi, c)
            else:
                phi = torch.where(c > math.cos(math.pi - hy.m), phi, c - math.sin(math.pi - hy.m) * hy.m)
            return c * hy.s, (onehot * phi + (1.0 - onehot) * c) * hy.s
        if self.kind == H.MAG:                                         # criterion.py:1245-1289
            xn = torch.norm(x, p=2, dim=1, keepdim=True).clamp(hy.l_a, hy.u_a)
            self.loss_g = torch.mean(1 / (hy.u_a ** 2) * xn + 1 / xn)
            c = c.clamp(-1 + 1e-7, 1 - 1e-7)
            am = (hy.u_margin - hy.l_margin) / (hy.u_a - hy.l_a) * (xn - hy.l_a) + hy.l_margin
            ctm = c * torch.cos(am) - torch.sqrt(1.0 - c ** 2 + 1e-9) * torch.sin(am)
            if hy.easy_margin:
                ctm = torch.where(c > 0, ctm, c)
            else:
                ctm = torch.where(c > torch.cos(math.pi - am), ctm, c - torch.sin(math.pi - am) * am)
            return c * hy.s, (onehot * ctm + (1.0 - onehot) * c) * hy.s
        c = c.clamp(-1, 1)
        ty = c[rows, labels].view(-1, 1)
        cm = ty * math.cos(hy.m) - torch.sqrt(1.0 - ty * ty) * math.sin(hy.m)
        mask = c > cm
        fin = torch.where(ty > math.cos(math.pi - hy.m), cm, ty - math.sin(math.pi - hy.m) * hy.m)
        with torch.no_grad():
            t = ty.mean() * hy.momentum + (1 - hy.momentum) * self.state.t
            self.state.t = float(t)
        origin = c.clone()
        z = torch.where(mask, c * (t + c), c)
        z = z.scatter(1, labels.view(-1, 1), fin)
        return origin * hy.s, z * hy.s


class FaceNet(nn.Module):
    """backbone + head, the *Net wrappers of criterion.py:109-135,199-230,303-325,589-617."""

    def __init__(self, kind, num_classes, hyper=None):
        super().__init__()
        from . import heads as H
        self.backbone = ResNet50()
        self.head = TorchHead(kind, FEATURE_DIM, num_classes, hyper or H.HeadHyper.default(kind))

    def forward(self, x, labels=None):
        feats = self.backbone(x)
        if self.training:
            assert labels is not None
            return self.head(feats, labels), feats
        return feats


def train_step(model: FaceNet, opt: torch.optim.Optimizer, images, labels):
    """model_utils.py:176-187 without AMP (autocast/GradScaler self-disable on CPU)."""
    model.train()
    (cos_s, logits), feats = model(images, labels)
    loss = F.cross_entropy(logits, labels)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.detach(), cos_s.detach(), logits.detach(), feats.detach()


def make_sgd(model, lr):
    return torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=5e-4)
