"""model.py counterpart: encoder + regression module + HIP decoder.

Mirrors `model.py:16-189` of the reference: `build_model` returns the same four handles
(segs_model, smpl_model, verts_model, projects_model), `build_full_model_from_saved_model` /
`build_full_model_for_predict` the same three.  The encoder (ENet stages 1-3,
`encoders/encoder_enet_simple.py:10-104`, or ResNet50) and the IEF / MLP regressor run on stock
PyTorch-ROCm ops (MIOpen convs, rocBLAS GEMMs), as the north star prescribes; the decoder is the
HIP path of this package (`decoder.SMPLDecoder`).  Layout is NCHW on the torch side; the reference's
(H, W, 3) `input_shape` is accepted and images may be passed NHWC or NCHW.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .decoder import SMPLDecoder
from .keras_smpl.concat_mean_param import concat_mean_param
from .keras_smpl.set_cam_params import load_mean_set_cam_params


class PReLU(nn.PReLU):
    """`PReLU(shared_axes=[1, 2])` (encoders/encoder_enet_simple.py:21): same parameters and state dict as
    `nn.PReLU(C)`; on a HIP device the forward / backward are the package's kernels (the stock backward is
    half of the train step: it materialises a per-element slope gradient before reducing it).  The slope starts
    at 0 as in Keras (`alpha_initializer='zeros'`), not at torch's 0.25."""

    def __init__(self, num_parameters=1, init=0.0, **kw):
        super().__init__(num_parameters, init=init, **kw)

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and self.weight.numel() == x.shape[1]
                and x.is_contiguous()):           # (NCHW planes; a channels_last tensor takes the stock op)
            from . import ops
            return ops.PReLUFn.apply(x.contiguous(), self.weight)
        return super().forward(x)


def _bn(ch, keras_momentum=0.99):
    # Keras `momentum` is the decay of the moving average; torch's is its complement. eps = 1e-3.
    return nn.BatchNorm2d(ch, eps=1e-3, momentum=1.0 - keras_momentum)


class _Bottleneck(nn.Module):
    """`bottleneck_enet`, encoders/encoder_enet_simple.py:27-80."""

    def __init__(self, cin, cout, internal_scale=4, asymmetric=0, dilated=0, downsample=False,
                 dropout_rate=0.1):
        super().__init__()
        internal = cout // internal_scale
        st = 2 if downsample else 1
        self.downsample, self.pad_ch = downsample, cout - cin
        self.reduce = nn.Sequential(nn.Conv2d(cin, internal, st, st, bias=False), _bn(internal, 0.1),
                                    PReLU(internal))
        if asymmetric:
            a = asymmetric
            conv = nn.Sequential(nn.Conv2d(internal, internal, (1, a), padding=(0, a // 2), bias=False),
                                 nn.Conv2d(internal, internal, (a, 1), padding=(a // 2, 0)))
        elif dilated:
            conv = nn.Conv2d(internal, internal, 3, padding=dilated, dilation=dilated)
        else:
            conv = nn.Conv2d(internal, internal, 3, padding=1)
        self.conv = nn.Sequential(conv, _bn(internal, 0.1), PReLU(internal))
        self.expand = nn.Sequential(nn.Conv2d(internal, cout, 1, bias=False), _bn(cout, 0.1),
                                    nn.Dropout2d(dropout_rate))
        self.act = PReLU(cout)

    def forward(self, x):
        # BatchNormalization (+ PReLU), and the BN -> dropout -> add -> PReLU tail, as single HIP ops when training
        from .ops import batch_norm_act, batch_norm_residual_act
        y = batch_norm_act(self.reduce[0](x), self.reduce[1], self.reduce[2])
        for m in list(self.conv)[:-2]:
            y = m(y)
        y = batch_norm_act(y, self.conv[-2], self.conv[-1])
        other = x
        if self.downsample:
            other = F.max_pool2d(other, 2)
            if self.pad_ch > 0:                       # zero-pad the feature maps (:66-73)
                other = F.pad(other, (0, 0, 0, 0, 0, self.pad_ch))
        return batch_norm_residual_act(self.expand[0](y), self.expand[1], self.expand[2], other, self.act)


class ENetEncoder(nn.Module):
    """`build_enet` (encoder_enet_simple.py:83-104): (N,3,256,256) -> (N,128,32,32)."""

    def __init__(self, dropout_rate=0.01):
        super().__init__()
        self.init_conv = nn.Conv2d(3, 13, 3, stride=2, padding=1)            # initial block (:10-14)
        self.init_bn, self.init_act = _bn(16, 0.1), PReLU(16)
        blocks = [_Bottleneck(16, 64, downsample=True, dropout_rate=dropout_rate)]
        blocks += [_Bottleneck(64, 64, dropout_rate=dropout_rate) for _ in range(4)]
        blocks += [_Bottleneck(64, 128, downsample=True)]
        for _ in range(2):
            blocks += [_Bottleneck(128, 128), _Bottleneck(128, 128, dilated=2),
                       _Bottleneck(128, 128, asymmetric=5), _Bottleneck(128, 128, dilated=4),
                       _Bottleneck(128, 128), _Bottleneck(128, 128, dilated=8),
                       _Bottleneck(128, 128, asymmetric=5), _Bottleneck(128, 128, dilated=16)]
        self.blocks = nn.Sequential(*blocks)

    def forward(self, x):
        from .ops import batch_norm_act
        x = torch.cat([self.init_conv(x), F.max_pool2d(x, 2)], dim=1)
        return self.blocks(batch_norm_act(x, self.init_bn, self.init_act))


class _ENetBackbone(nn.Module):
    """ENet + the three conv/BN/pool stages of model.py:38-54 -> (N, 2048)."""

    def __init__(self):
        super().__init__()
        self.enet = ENetEncoder()

        def stage(cin, cout, pool):
            return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(inplace=True), _bn(cout),
                                 nn.MaxPool2d(pool))
        self.head = nn.Sequential(stage(128, 256, 4), stage(256, 512, 2), stage(512, 2048, 4))

    def forward(self, x):
        if x.shape[2] != 256 or x.shape[3] != 256:
            raise RuntimeError("the ENet branch needs 256x256 inputs (model.py:40-54 ends in Reshape((2048,)))")
        return self.head(self.enet(x)).flatten(1)


class _ResBlock(nn.Module):
    def __init__(self, cin, mid, stride, project):
        super().__init__()
        self.c1 = nn.Sequential(nn.Conv2d(cin, mid, 1, stride), _bn(mid), nn.ReLU(inplace=True))
        self.c2 = nn.Sequential(nn.Conv2d(mid, mid, 3, padding=1), _bn(mid), nn.ReLU(inplace=True))
        self.c3 = nn.Sequential(nn.Conv2d(mid, mid * 4, 1), _bn(mid * 4))
        self.proj = nn.Sequential(nn.Conv2d(cin, mid * 4, 1, stride), _bn(mid * 4)) if project else None

    def forward(self, x):
        s = x if self.proj is None else self.proj(x)
        return F.relu(self.c3(self.c2(self.c1(x))) + s)


class _ResNet50Backbone(nn.Module):
    """keras.applications.resnet50.ResNet50(include_top=False, weights=None) of Keras 2.1 (7x7 average
    pool kept) followed by Reshape((2048,)) (model.py:56-60)."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3), _bn(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
        layers, cin = [], 64
        for mid, n, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
            for i in range(n):
                layers.append(_ResBlock(cin, mid, stride if i == 0 else 1, i == 0))
                cin = mid * 4
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        f = F.avg_pool2d(self.layers(self.stem(x)), 7)
        if f.shape[2] != 1 or f.shape[3] != 1:
            raise RuntimeError("ResNet50 branch: Reshape((2048,)) needs a 1x1 feature map (224..287 px inputs)")
        return f.flatten(1)


class SMPLRegressor(nn.Module):
    """Backbone + regression module (model.py:36-105): images -> final_param (N, 86).

    This is the reference's `smpl_model` (the only part with trainable weights, train.py:302-315).
    """

    def __init__(self, output_wh, encoder_architecture="resnet50", use_IEF=False, scaledown=0.005):
        super().__init__()
        if encoder_architecture == "enet":
            self.backbone = _ENetBackbone()
        elif encoder_architecture == "resnet50":
            self.backbone = _ResNet50Backbone()
        else:
            raise ValueError("encoder_architecture must be 'enet' or 'resnet50' (model.py:38,56)")
        self.output_wh, self.use_IEF, self.scaledown = output_wh, bool(use_IEF), float(scaledown)
        if use_IEF:                                            # shared across the 3 iterations (:66-68)
            self.IEF_layer_1 = nn.Linear(2048 + 86, 1024)
            self.IEF_layer_2 = nn.Linear(1024, 1024)
            self.IEF_layer_3 = nn.Linear(1024, 86)
        else:
            self.mlp = nn.Sequential(nn.Linear(2048, 2048), nn.ReLU(inplace=True),
                                     nn.Linear(2048, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 86))

    def forward(self, images):
        if images.dim() == 4 and images.shape[1] != 3 and images.shape[3] == 3:
            images = images.permute(0, 3, 1, 2)               # NHWC (Keras) -> NCHW
        feats = self.backbone(images.contiguous())
        if not self.use_IEF:
            return load_mean_set_cam_params(self.mlp(feats) * self.scaledown, self.output_wh)   # :100-105
        state = concat_mean_param(feats, self.output_wh)                                         # :70-71
        param = state[:, 2048:]
        for _ in range(3):                                                                       # :77-97
            delta = self.IEF_layer_3(F.relu(self.IEF_layer_2(F.relu(self.IEF_layer_1(state)))))
            param = param + delta * self.scaledown
            state = torch.cat([feats, param], dim=1)
        return param


class FullModel(nn.Module):
    """smpl_model + decoder.  `output` selects what forward returns: 'segs' (softmaxed (N, W*W, 32),
    model.py:119-120), 'segs_raw' ((N,W,W,32), the predict path model.py:176-184), 'smpl', 'verts',
    'projects', or 'all' (dict)."""

    def __init__(self, smpl_model, decoder: SMPLDecoder, output="segs"):
        super().__init__()
        self.smpl_model, self.decoder, self.output = smpl_model, decoder, output

    def forward(self, images):
        param = self.smpl_model(images)
        if self.output == "smpl":
            return param
        out = self.decoder(param)
        out["smpl"] = param
        if self.output == "all":
            return out
        if self.output == "verts":
            return out["verts"]
        if self.output == "projects":
            return out["projects"]
        if self.output == "segs_raw":
            return out["seg"]
        if self.output == "silhs":                     # train_stage2_silhouette.py:85-86
            sil = out["silhouette"]
            return torch.softmax(sil.reshape(sil.shape[0], -1, sil.shape[-1]), dim=-1)
        seg = out["seg"]
        return torch.softmax(seg.reshape(seg.shape[0], -1, seg.shape[-1]), dim=-1)


def build_model(train_batch_size, input_shape, smpl_path, output_wh, num_classes,
                encoder_architecture="resnet50", use_IEF=False, vertex_sampling=None, scaledown=0.005):
    """`build_model`, model.py:16-129.  Returns (segs_model, smpl_model, verts_model, projects_model);
    the four share one encoder/regressor and one decoder, as the Keras models share one graph."""
    if num_classes != 32:
        raise ValueError("the decoder produces 32 classes (31 parts + background)")
    smpl_model = SMPLRegressor(output_wh, encoder_architecture, use_IEF, scaledown)
    decoder = SMPLDecoder(smpl_path, img_wh=output_wh, vertex_sampling=vertex_sampling)
    return (FullModel(smpl_model, decoder, "segs"), smpl_model, FullModel(smpl_model, decoder, "verts"),
            FullModel(smpl_model, decoder, "projects"))


class EmbeddedSMPLParams(nn.Module):
    """The "encoder" of decoder_loss_debugging.py:69-77: one learnable 86-vector per sample index
    (`Embedding(25, 86)`, Keras' uniform(-0.05, 0.05) initialiser) + `load_mean_set_cam_params`."""

    def __init__(self, output_wh, num_embeddings=25):
        super().__init__()
        self.output_wh = output_wh
        self.table = nn.Embedding(num_embeddings, 86)
        nn.init.uniform_(self.table.weight, -0.05, 0.05)

    def forward(self, index_inputs):
        idx = index_inputs.reshape(index_inputs.shape[0]).long()
        return load_mean_set_cam_params(self.table(idx), self.output_wh)


def build_debug_model(batch_size, smpl_path, output_img_wh, num_classes, vertex_sampling=None):
    """`build_debug_model`, decoder_loss_debugging.py:69-100: the decoder driven by a table of learnable SMPL
    parameters instead of an image encoder - fitting it to target segmentations exercises nothing but the
    decoder's forward and backward.  Returns (segs_model, smpl_model, verts_model, projects_model) on index inputs."""
    if num_classes != 32:
        raise ValueError("the decoder produces 32 classes (31 parts + background)")
    smpl_model = EmbeddedSMPLParams(output_img_wh)
    decoder = SMPLDecoder(smpl_path, img_wh=output_img_wh, vertex_sampling=vertex_sampling)
    return (FullModel(smpl_model, decoder, "segs"), smpl_model, FullModel(smpl_model, decoder, "verts"),
            FullModel(smpl_model, decoder, "projects"))


def build_full_model_from_saved_model_stage2(smpl_model, segs_output_wh, silhs_output_wh, smpl_path, batch_size,
                                             num_classes_segs=32, num_classes_silhs=2):
    """`build_full_model_from_saved_model` of train_stage2_silhouette.py:72-104: around a saved encoder, the part
    segmentation at `segs_output_wh` and the silhouette at its own `silhs_output_wh` from ONE decoder pass.
    Returns (verts_model, projects_model, silhouettes_model, segs_model)."""
    if num_classes_segs != 32 or num_classes_silhs != 2:
        raise ValueError("the decoder produces 32 part classes and 2 silhouette classes")
    decoder = SMPLDecoder(smpl_path, img_wh=segs_output_wh, with_silhouette=True, silh_wh=silhs_output_wh)
    return (FullModel(smpl_model, decoder, "verts"), FullModel(smpl_model, decoder, "projects"),
            FullModel(smpl_model, decoder, "silhs"), FullModel(smpl_model, decoder, "segs"))


def build_full_model_from_saved_model(smpl_model, output_wh, smpl_path, batch_size, num_classes):
    """model.py:132-161: (verts_model, projects_model, segs_model) around a saved encoder."""
    decoder = SMPLDecoder(smpl_path, img_wh=output_wh)
    return (FullModel(smpl_model, decoder, "verts"), FullModel(smpl_model, decoder, "projects"),
            FullModel(smpl_model, decoder, "segs"))


def build_full_model_for_predict(smpl_model, output_wh, smpl_path, batch_size=1):
    """model.py:164-189: as above but raw (N,W,W,32) scores, no reshape/softmax."""
    decoder = SMPLDecoder(smpl_path, img_wh=output_wh)
    return (FullModel(smpl_model, decoder, "verts"), FullModel(smpl_model, decoder, "projects"),
            FullModel(smpl_model, decoder, "segs_raw"))
