"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the MaGGIe matting hot path.

A *functional* PyTorch-CPU restatement that works directly on a reference-compatible ``state_dict``
(``Dict[str, Tensor]`` with the reference's key names) instead of on nn.Modules. Every function cites
the reference file:line (relative to /root/reference/) it follows. The sparse refinement head calls into
the third-party `spconv` CUDA library in the reference (un-pinned, absent here) and `compute_unknown`
calls OpenCV (un-pinned, absent here): both are restated from their published semantics in
``oracle/region.py`` and below => **parity unpinned** for rows U1 and S0-S9 of SURVEY.md section 8;
all other rows are pinned against the reference's own modules by tests/golden (see make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this module.
The product (maggie_amd/) never imports it.
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import region

LRELU = 0.2


# ----------------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------------

def sn_weight(sd, p):
    """SpectralNorm._update_u_v -- maggie/network/module/spectral_norm.py:22-35.
    One power iteration on every call (train AND eval), in place on weight_u / weight_v, no grad through
    u, v; returns weight_bar / sigma."""
    w = sd[p + '.weight_bar']
    u = sd[p + '.weight_u']
    v = sd[p + '.weight_v']
    h = w.shape[0]
    with torch.no_grad():
        wm = w.detach().reshape(h, -1)
        nv = torch.mv(wm.t(), u.detach())
        nv = nv / (nv.norm() + 1e-12)
        nu = torch.mv(wm, nv)
        nu = nu / (nu.norm() + 1e-12)
    # the reference rebinds `.data` (a NEW tensor each call), so graphs of earlier calls keep their own u, v
    sd[p + '.weight_u'] = nu
    sd[p + '.weight_v'] = nv
    sigma = nu.dot(w.reshape(h, -1).mv(nv))
    return w / sigma


def bn(sd, p, x, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm{1,2}d forward (batch statistics + running-stat update in training)."""
    if training and (p + '.num_batches_tracked') in sd:
        sd[p + '.num_batches_tracked'] += 1
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        training, momentum, eps)


def sn_conv(sd, p, x, stride=1, padding=0, dilation=1):
    return F.conv2d(x, sn_weight(sd, p + '.module'), None, stride, padding, dilation)


def linear(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd[p + '.bias'])


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def resize_any_shape(x, scale_factor, use_max_pool=False, use_avg_pool_binary=False):
    """maggie/utils/utils.py:7-25 (only the two pooling modes are on the path)."""
    shape = x.shape
    dtype = x.dtype
    x = x.reshape(-1, shape[-3], *shape[-2:]).float()
    stride = int(1 / scale_factor)
    if use_max_pool:
        x = F.max_pool2d(x, kernel_size=stride, stride=stride)
    elif use_avg_pool_binary:
        x = F.avg_pool2d(x, kernel_size=stride, stride=stride)
        x = (x > 0.0).float()
    else:
        raise NotImplementedError
    return x.reshape(*shape[:-2], *x.shape[-2:]).to(dtype)


def compute_unknown(masks, k_size=30, is_train=False):
    """maggie/utils/utils.py:28-55; returns a uint8 tensor shaped like `masks`."""
    out = region.compute_unknown(masks.detach().float().cpu().numpy(), k_size, is_train)      # .float(): also usable under CPU bf16 autocast (bf16 yardstick runs)
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------------------------------
# encoder  (maggie/network/encoder/resnet.py)
# ----------------------------------------------------------------------------------------------------

def enc_block(sd, p, x, stride, training):
    """BasicBlock.forward -- encoder/resnet.py:23-39; downsample = AvgPool2d(2,stride)+SN 1x1+BN (:111-116)."""
    out = sn_conv(sd, p + '.conv1', x, stride, 1)
    out = F.relu(bn(sd, p + '.bn1', out, training))
    out = sn_conv(sd, p + '.conv2', out, 1, 1)
    out = bn(sd, p + '.bn2', out, training)
    idt = x
    if (p + '.downsample.1.module.weight_bar') in sd:
        idt = F.avg_pool2d(x, 2, stride)
        idt = bn(sd, p + '.downsample.2', sn_conv(sd, p + '.downsample.1', idt), training)
    elif (p + '.downsample.0.module.weight_bar') in sd:
        idt = bn(sd, p + '.downsample.1', sn_conv(sd, p + '.downsample.0', x, stride), training)
    return F.relu(out + idt)


def enc_layer(sd, p, x, n_blocks, stride, training):
    for i in range(n_blocks):
        x = enc_block(sd, '%s.%d' % (p, i), x, stride if i == 0 else 1, training)
    return x


def enc_shortcut(sd, p, x, training):
    """_make_shortcut -- encoder/resnet.py:167-175: (SN conv3x3 -> ReLU -> BN) x 2 (ReLU BEFORE BN)."""
    x = bn(sd, p + '.2', F.relu(sn_conv(sd, p + '.0', x, 1, 1)), training)
    x = bn(sd, p + '.5', F.relu(sn_conv(sd, p + '.3', x, 1, 1)), training)
    return x


def mask_id_embedding(sd, p, masks):
    """ResMaskEmbedShortCut_D.forward -- encoder/resnet.py:214-225: (N,n,h,w) float masks -> (N,3,h,w)."""
    ids = torch.arange(1, masks.shape[1] + 1)[None, :, None, None]
    m = (masks * ids).long()
    emb = F.embedding(m, sd[p + '.mask_embed_layer.weight'])          # N,n,h,w,E
    on = (m > 0).float().unsqueeze(-1)
    emb = (emb * on).sum(1) / (on.sum(1) + 1e-6)
    return emb.permute(0, 3, 1, 2)


def encoder(sd, p, inp, training, layers=(3, 4, 4, 2)):
    """res_shortcut_embed_29 forward -- encoder/resnet.py:211-229 then :177-200.
    inp: (N, 3 + num_mask, h, w). Returns (out_os32, dict(shortcut=(fea1..fea5), image=...))."""
    img = inp[:, :3]
    x = torch.cat([img, mask_id_embedding(sd, p, inp[:, 3:])], 1)
    out = F.relu(bn(sd, p + '.bn1', sn_conv(sd, p + '.conv1', x, 2, 1), training))
    x1 = F.relu(bn(sd, p + '.bn2', sn_conv(sd, p + '.conv2', out, 1, 1), training))
    out = F.relu(bn(sd, p + '.bn3', sn_conv(sd, p + '.conv3', x1, 2, 1), training))
    x2 = enc_layer(sd, p + '.layer1', out, layers[0], 1, training)
    x3 = enc_layer(sd, p + '.layer2', x2, layers[1], 2, training)
    x4 = enc_layer(sd, p + '.layer3', x3, layers[2], 2, training)
    out = enc_layer(sd, p + '.layer_bottleneck', x4, layers[3], 2, training)
    fea1 = enc_shortcut(sd, p + '.shortcut.0', x, training)
    fea2 = enc_shortcut(sd, p + '.shortcut.1', x1, training)
    fea3 = enc_shortcut(sd, p + '.shortcut.2', x2, training)
    fea4 = enc_shortcut(sd, p + '.shortcut.3', x3, training)
    fea5 = enc_shortcut(sd, p + '.shortcut.4', x4, training)
    return out, {'shortcut': (fea1, fea2, fea3, fea4, fea5), 'image': x[:, :3], 'backbone_feat': (x2, x3, x4, out)}


# ----------------------------------------------------------------------------------------------------
# ASPP  (maggie/network/module/aspp.py:34-56)
# ----------------------------------------------------------------------------------------------------

def aspp(sd, p, x, training):
    def br(name, inp, dil, k):
        w = sd['%s.%s.weight' % (p, name)]
        y = F.conv2d(inp, w, None, 1, dil if k == 3 else 0, dil)
        return F.relu(bn(sd, '%s.%s_bn' % (p, name), y, training))
    x1 = br('aspp1', x, 1, 1)
    x2 = br('aspp2', x, 2, 3)
    x3 = br('aspp3', x, 4, 3)
    x4 = br('aspp4', x, 8, 3)
    x5 = br('aspp5', F.adaptive_avg_pool2d(x, 1), 1, 1)
    x5 = F.interpolate(x5, size=(x.shape[2], x.shape[3]), mode='nearest')
    y = torch.cat((x1, x2, x3, x4, x5), 1)
    y = F.conv2d(y, sd[p + '.conv2.weight'])
    return F.relu(bn(sd, p + '.bn2', y, training))


# ----------------------------------------------------------------------------------------------------
# decoder dense part (maggie/network/decoder/resnet.py:9-45, resnet_inst_matt_spconv.py:368-388)
# ----------------------------------------------------------------------------------------------------

def dec_block(sd, p, x, stride, training):
    if stride > 1:
        out = F.conv_transpose2d(x, sn_weight(sd, p + '.conv1.module'), None, 2, 1)
    else:
        out = sn_conv(sd, p + '.conv1', x, 1, 1)
    out = F.leaky_relu(bn(sd, p + '.bn1', out, training), LRELU)
    out = bn(sd, p + '.bn2', sn_conv(sd, p + '.conv2', out, 1, 1), training)
    idt = x
    if (p + '.upsample.1.module.weight_bar') in sd:
        idt = F.interpolate(x, scale_factor=2, mode='nearest')
        idt = bn(sd, p + '.upsample.2', sn_conv(sd, p + '.upsample.1', idt), training)
    elif (p + '.upsample.0.module.weight_bar') in sd:
        idt = bn(sd, p + '.upsample.1', sn_conv(sd, p + '.upsample.0', x), training)
    return F.leaky_relu(out + idt, LRELU)


def dec_layer(sd, p, x, n_blocks, training):
    for i in range(n_blocks):
        x = dec_block(sd, '%s.%d' % (p, i), x, 2 if i == 0 else 1, training)
    return x


def os32_to_os8(sd, p, x, mid_fea, training):
    fea1, fea2, fea3, fea4, fea5 = mid_fea['shortcut']
    x = dec_layer(sd, p + '.layer1', x, 2, training) + fea5
    x = dec_layer(sd, p + '.layer2', x, 3, training) + fea4
    return x


# ----------------------------------------------------------------------------------------------------
# attention blocks (maggie/network/module/mask_attention.py:9-206)
# ----------------------------------------------------------------------------------------------------

def _mha(sd, p, q, k, v, key_padding_mask=None, attn_mask=None):
    d = q.shape[-1]
    return F.multi_head_attention_forward(
        q, k, v, d, 1, sd[p + '.in_proj_weight'], sd[p + '.in_proj_bias'], None, None, False, 0.0,
        sd[p + '.out_proj.weight'], sd[p + '.out_proj.bias'], training=False,
        key_padding_mask=key_padding_mask, need_weights=True, attn_mask=attn_mask)


def cross_attention(sd, p, tgt, memory, pos, query_pos, memory_key_padding_mask=None, memory_mask=None):
    """CrossAttentionLayer.forward_post -- mask_attention.py:90-113 (dropout 0, post-norm)."""
    if torch.isnan(tgt).any():
        raise ValueError("Mask is empty")
    q = tgt if query_pos is None else tgt + query_pos
    k = memory if pos is None else memory + pos
    tgt2, att = _mha(sd, p + '.multihead_attn', q, k, memory, memory_key_padding_mask, memory_mask)
    return layer_norm(sd, p + '.norm', tgt + tgt2), att


def self_attention(sd, p, tgt, query_pos, key_padding_mask):
    """SelfAttentionLayer.forward_post -- mask_attention.py:31-41."""
    q = tgt + query_pos
    tgt2 = _mha(sd, p + '.self_attn', q, q, tgt, key_padding_mask)[0]
    return layer_norm(sd, p + '.norm', tgt + tgt2)


def ffn(sd, p, tgt, drop_p=0.0, training=False):
    """FFNLayer.forward_post -- mask_attention.py:168-172."""
    h = F.dropout(F.relu(linear(sd, p + '.linear1', tgt)), drop_p, training)
    tgt2 = F.dropout(linear(sd, p + '.linear2', h), drop_p, training)
    return layer_norm(sd, p + '.norm', tgt + tgt2)


# ----------------------------------------------------------------------------------------------------
# ConvGRU (maggie/network/module/conv_gru.py:4-69)
# ----------------------------------------------------------------------------------------------------

def conv_gru_frame(sd, p, x, h):
    C = x.shape[1]
    rz = torch.sigmoid(F.conv2d(torch.cat([x, h], 1), sd[p + '.ih.0.weight'], sd[p + '.ih.0.bias'], 1, 1))
    r, z = rz.split(C, dim=1)
    c = torch.tanh(F.conv2d(torch.cat([x, r * h], 1), sd[p + '.hh.0.weight'], sd[p + '.hh.0.bias'], 1, 1))
    return (1 - z) * h + z * c


def conv_gru_series(sd, p, x, h):
    if h is None:
        h = torch.zeros((x.size(0), x.size(-3), x.size(-2), x.size(-1)), dtype=x.dtype)
    outs = []
    for t in range(x.shape[1]):
        h = conv_gru_frame(sd, p, x[:, t], h)
        outs.append(h)
    return torch.stack(outs, 1)


def conv_gru_propagate(sd, p, feat, n_f, prev_h_state=None, temp_method='bi'):
    """ConvGRU.propagate_features -- conv_gru.py:50-69. feat: (b, n_f, C, h, w)."""
    if temp_method == 'none':
        outs = [conv_gru_frame(sd, p, feat[:, j], torch.zeros_like(feat[:, j])) for j in range(n_f)]
        return torch.stack(outs, 1), outs[-1]
    fwd = conv_gru_series(sd, p, feat, prev_h_state)
    hidden = fwd
    if temp_method == 'bi':
        bwd = conv_gru_series(sd, p, torch.flip(feat[:, :-1], dims=(1,)), hidden[:, -1])
        bwd = torch.flip(bwd, dims=(1,))
        out = torch.cat([(fwd[:, :-1] + bwd) / 2, fwd[:, -1:]], 1)
    else:
        out = fwd
    return out, hidden


# ----------------------------------------------------------------------------------------------------
# InstanceMatteDecoder (maggie/network/module/instance_matte_decoder.py:112-307)
# ----------------------------------------------------------------------------------------------------

def imd(sd, p, ori_feat, mask, training, gt_mask=None, aggregate_mem_fn=None, n_block=2, max_inst=10):
    """use_mask_atten is always False at the target configs; atten_stride=1, use_id_pe=True,
    use_temp_pe=False. ori_feat: (b*n_f, 128, h, w); mask: (b, n_f, n_i, H, W) float."""
    feat = ori_feat
    scale_factor = feat.shape[-1] * 1.0 / mask.shape[-1] * 1.0
    mask = resize_any_shape(mask, scale_factor, use_avg_pool_binary=True)
    b, n_f = mask.shape[:2]
    h, w = feat.shape[-2:]
    C = feat.shape[1]
    feat = feat.reshape(b, n_f, 1, C, h * w)

    idw = sd[p + '.id_embedding.weight']
    ids = torch.arange(1, mask.shape[2] + 1)[None, None, :, None, None]
    id_feat_pos = (mask * ids).max(2)[0]
    id_feat_pos = F.embedding(id_feat_pos.long(), idw)                 # b, n_f, h, w, c
    feat_pos = id_feat_pos.permute(0, 1, 4, 2, 3).reshape(b, n_f, 1, -1, h * w)

    tokens = sd[p + '.query_feat.weight'][None].repeat(b, 1, 1)        # b, 10, c
    token_pos = F.embedding(torch.arange(1, max_inst + 1), idw)[None].repeat(b, 1, 1)

    feat = feat.permute(4, 2, 1, 0, 3).reshape(h * w * n_f, b, -1)
    feat_pos = feat_pos.permute(4, 2, 1, 0, 3).reshape(h * w * n_f, b, -1)
    feat = linear(sd, p + '.feat_proj.layers.0', feat)
    n_i = max_inst
    tokens = tokens.permute(1, 0, 2)
    token_pos = token_pos.permute(1, 0, 2)

    guidance_mask = None
    if training:
        gm = resize_any_shape(gt_mask, scale_factor, use_max_pool=True)     # bool in, bool out
        gm = gm.permute(1, 0, 2, 3, 4).reshape(n_f * b, -1, h * w)
        if gm.shape[1] < n_i:
            gm = torch.cat([gm.float(), torch.zeros((n_f * b, n_i - gm.shape[1], h * w))], 1)
        gm = gm > 0
        guidance_mask = gm.reshape(n_f, b, n_i, -1).permute(1, 2, 3, 0).flatten(2, 3)

    def atten_loss(att):
        vals = (guidance_mask * att).sum(2)
        gt = torch.ones_like(vals)
        gt[guidance_mask.sum(2) == 0] = 0
        return (gt - vals).sum() / (n_f * b)

    max_loss = 0
    valid_tokens = mask.sum((1, 3, 4)) > 0
    if valid_tokens.shape[1] < n_i:
        valid_tokens = torch.cat([valid_tokens, torch.zeros((b, n_i - valid_tokens.shape[1])).bool()], 1)
    token_padding_mask = ~valid_tokens

    for i in range(n_block):
        tokens, att = cross_attention(sd, '%s.token_feat_ca_layers.%d' % (p, i), tokens, feat, feat_pos, token_pos)
        if training:
            max_loss = max_loss + atten_loss(att)
        tokens = ffn(sd, '%s.mlp_layers.%d' % (p, i), tokens)
        tokens = self_attention(sd, '%s.sa_layers.%d' % (p, i), tokens, token_pos, token_padding_mask)
        feat, _ = cross_attention(sd, '%s.feat_token_ca_layers.%d' % (p, i), feat, tokens, token_pos, feat_pos,
                                  memory_key_padding_mask=token_padding_mask)
    tokens, att = cross_attention(sd, p + '.final_token_feat_ca', tokens, feat, feat_pos, token_pos)
    if training:
        max_loss = max_loss + atten_loss(att)
    max_loss = max_loss / (n_block + 1)

    feat = feat.reshape(h, w, n_f, b, -1).permute(3, 2, 4, 0, 1).reshape(b * n_f, -1, h, w)

    def smooth(x):
        x = F.conv2d(x, sd[p + '.conv.0.weight'], None, 1, 1)
        x = F.leaky_relu(bn(sd, p + '.conv.1', x, training), LRELU)
        x = F.conv2d(x, sd[p + '.conv.3.weight'])
        return F.leaky_relu(bn(sd, p + '.conv.4', x, training), LRELU)

    hidden_state = None
    if aggregate_mem_fn is not None:
        no_temp = feat
        feat, hidden_state = aggregate_mem_fn(feat.reshape(b, n_f, -1, h, w))
        feat = feat.flatten(0, 1)
        out_feat = smooth(no_temp)
        feat = smooth(feat)
    else:
        feat = smooth(feat)
        out_feat = feat

    tokens = linear(sd, p + '.final_mlp.layers.0', tokens)
    tokens = tokens.reshape(n_i, b, -1).permute(1, 0, 2)
    tokens = layer_norm(sd, p + '.decoder_norm', tokens)
    output_mask = torch.einsum('bqc,btchw->btqhw', tokens, feat.reshape(b, n_f, -1, h, w)).flatten(0, 1)
    return output_mask, out_feat, tokens, max_loss, hidden_state


# ----------------------------------------------------------------------------------------------------
# sparse refinement head  (resnet_inst_matt_spconv.py:161-270) -- spconv semantics restated, unpinned
# ----------------------------------------------------------------------------------------------------

def _gather_conv(feat, nbr, w_krsc, bias=None):
    """out[r] = sum_k W[:, k, :] @ feat[nbr[r, k]]  (nbr == -1 skipped). w_krsc: (Cout, kh, kw, Cin)."""
    cout = w_krsc.shape[0]
    K = nbr.shape[1]
    wk = w_krsc.reshape(cout, K, -1)
    out = feat.new_zeros((nbr.shape[0], cout))
    nbr_t = torch.from_numpy(nbr).long()
    for k in range(K):
        idx = nbr_t[:, k]
        ok = idx >= 0
        if ok.any():
            out = out.index_add(0, torch.nonzero(ok)[:, 0], feat[idx[ok]] @ wk[:, k, :].t())
    if bias is not None:
        out = out + bias
    return out


def _rows_linear(feat, w_krsc, bias=None):
    """SubMConv2d with kernel_size=1 (any padding): per-site linear map."""
    out = feat @ w_krsc.reshape(w_krsc.shape[0], -1).t()
    return out if bias is None else out + bias


def _bn1d(sd, p, x, training):
    if x.shape[0] == 0:
        return x
    return bn(sd, p, x, training)


def predict_details(sd, p, os8_feat, image, roi_masks, n_i, inst_guidance_os8, dense_features, training,
                    drop_p=0.1):
    """predict_details -- resnet_inst_matt_spconv.py:196-270.
    os8_feat (N,64,H/8,W/8); image (N,3,H,W); roi_masks (N,n_i,H,W) uint8; inst_guidance_os8 (N,10,64).
    Returns dense (N*n_i,1,H/4,W/4), (N*n_i,1,H,W) logits with -99 outside the active sites."""
    N, _, H, W = roi_masks.shape
    roi = (roi_masks.reshape(N * n_i, H, W) > 0).numpy()
    a1, a2, a4, a8 = region.active_pyramid(roi)
    c1, c2, c4, c8 = [region.coords_of(a) for a in (a1, a2, a4, a8)]
    fea1, fea2, fea3 = dense_features

    def frames(c):
        return torch.from_numpy(c[:, 0] // n_i).long()

    def yx(c):
        return torch.from_numpy(c[:, 1]).long(), torch.from_numpy(c[:, 2]).long()

    def gather_dense(dense, c):
        y, x = yx(c)
        return dense.permute(0, 2, 3, 1)[frames(c), y, x]

    # S2: OS8 gather + instance guidance (:221-232)
    x = gather_dense(os8_feat, c8)
    inst = torch.from_numpy(c8[:, 0] % n_i).long()
    guidance = inst_guidance_os8[frames(c8), inst]
    x = ffn(sd, p + '.inst_spec_layer', x * guidance, drop_p, training)

    # S3: layer3 (:69-74,238)
    x = _gather_conv(x, region.inverse_neighbors(a4, a8), sd[p + '.layer3.0.weight'])
    x = F.leaky_relu(_bn1d(sd, p + '.layer3.1', x, training), LRELU)
    nbr4 = region.subm_neighbors(a4)
    x = _gather_conv(x, nbr4, sd[p + '.layer3.3.weight'])

    # S4: instance_spec_guidance (:172-194) + guidance_layer (:76-82)
    detail = gather_dense(fea3, c4)
    g = _rows_linear(torch.cat([detail, x], 1), sd[p + '.guidance_layer.0.weight'])
    g = F.leaky_relu(_bn1d(sd, p + '.guidance_layer.1', g, training), LRELU)
    g = torch.sigmoid(_gather_conv(g, nbr4, sd[p + '.guidance_layer.3.weight'], sd[p + '.guidance_layer.3.bias']))
    x = detail * g

    # S5: layer3_smooth (:84-88), refine_OS4 (:118-123), densify (:247-251)
    x = _rows_linear(x, sd[p + '.layer3_smooth.0.weight'], sd[p + '.layer3_smooth.0.bias'])
    x = _bn1d(sd, p + '.layer3_smooth.2', F.relu(x), training)
    o4 = _gather_conv(x, nbr4, sd[p + '.refine_OS4.0.weight'])
    o4 = F.leaky_relu(_bn1d(sd, p + '.refine_OS4.1', o4, training), LRELU)
    o4 = _gather_conv(o4, nbr4, sd[p + '.refine_OS4.3.weight'], sd[p + '.refine_OS4.3.bias'])
    x_os4 = torch.full((N * n_i, 1, H // 4, W // 4), -99.0)
    y4, x4 = yx(c4)
    x_os4 = x_os4.index_put((torch.from_numpy(c4[:, 0]).long(), torch.zeros_like(y4), y4, x4), o4[:, 0])

    # S6: layer4 (:91-96), fea2, layer4_smooth (:98-102)
    x = _gather_conv(x, region.inverse_neighbors(a2, a4), sd[p + '.layer4.0.weight'])
    x = F.leaky_relu(_bn1d(sd, p + '.layer4.1', x, training), LRELU)
    x = _rows_linear(x, sd[p + '.layer4.3.weight'])                 # SubMConv2d(k=1, padding=1) == per-site linear
    x = torch.cat([gather_dense(fea2, c2), x], 1)
    x = _rows_linear(x, sd[p + '.layer4_smooth.0.weight'], sd[p + '.layer4_smooth.0.bias'])
    x = _bn1d(sd, p + '.layer4_smooth.2', F.relu(x), training)

    # S7: layer5 (:105-110), fea1, layer5_smooth (:112-116), refine_OS1 (:125-130), densify (:264-268)
    x = _gather_conv(x, region.inverse_neighbors(a1, a2), sd[p + '.layer5.0.weight'])
    x = F.leaky_relu(_bn1d(sd, p + '.layer5.1', x, training), LRELU)
    nbr1 = region.subm_neighbors(a1)
    x = _gather_conv(x, nbr1, sd[p + '.layer5.3.weight'])
    x = torch.cat([gather_dense(fea1, c1), x], 1)
    x = _rows_linear(x, sd[p + '.layer5_smooth.0.weight'], sd[p + '.layer5_smooth.0.bias'])
    x = _bn1d(sd, p + '.layer5_smooth.2', F.relu(x), training)
    o1 = _gather_conv(x, nbr1, sd[p + '.refine_OS1.0.weight'])
    o1 = F.leaky_relu(_bn1d(sd, p + '.refine_OS1.1', o1, training), LRELU)
    o1 = _gather_conv(o1, nbr1, sd[p + '.refine_OS1.3.weight'], sd[p + '.refine_OS1.3.bias'])
    x_os1 = torch.full((N * n_i, 1, H, W), -99.0)
    y1, x1 = yx(c1)
    x_os1 = x_os1.index_put((torch.from_numpy(c1[:, 0]).long(), torch.zeros_like(y1), y1, x1), o1[:, 0])
    return x_os4, x_os1


def dec_fuse(pred, detail_mask, training):
    """ResShortCut_InstMattSpconv_Dec.fuse -- resnet_inst_matt_spconv.py:272-290."""
    a1, a4, a8 = pred['alpha_os1'], pred['alpha_os4'], pred['alpha_os8']
    alpha = a8
    w4 = compute_unknown(alpha, 27, training) * detail_mask
    w4 = (w4 > 0).type(alpha.dtype)
    alpha = a4 * w4 + alpha * (1 - w4)
    w1 = compute_unknown(alpha, 15, training) * detail_mask
    w1 = (w1 > 0).type(alpha.dtype)
    alpha = a1 * w1 + alpha * (1 - w1)
    return alpha, w4, w1


def process_os4_os1(sd, p, x, b, n_f, fea1, fea2, fea3, image, x_os8, queries, guided_mask_os8, unknown_os8, training):
    """process_os4_os1 -- resnet_inst_matt_spconv.py:346-366."""
    if unknown_os8.max() == 0 and training:
        unknown_os8[:, :, 200:250, 200:250] = 1
    if unknown_os8.sum() > 0 or training:
        queries = queries[:, None].expand(-1, n_f, -1, -1).reshape(b * n_f, *queries.shape[1:])
        n_i = guided_mask_os8.shape[1]
        x_os4, x_os1 = predict_details(sd, p, x, image, unknown_os8, n_i, queries, [fea1, fea2, fea3], training)
        x_os4 = x_os4.reshape(b * n_f, n_i, *x_os4.shape[-2:])
        x_os1 = x_os1.reshape(b * n_f, n_i, *x_os1.shape[-2:])
        x_os4 = F.interpolate(x_os4, scale_factor=4.0, mode='bilinear', align_corners=False)
        x_os4 = (torch.tanh(x_os4) + 1.0) / 2.0
        x_os1 = (torch.tanh(x_os1) + 1.0) / 2.0
    else:
        x_os4 = torch.zeros((b * n_f, x_os8.shape[1], image.shape[2], image.shape[3]))
        x_os1 = torch.zeros_like(x_os4)
    return x_os4, x_os1


def _dec_prologue(sd, p, x, mid_fea, b, n_f, n_i, masks, gt_alphas, training):
    masks = masks.reshape(b, n_f, n_i, masks.shape[2], masks.shape[3])
    valid_masks = masks.flatten(0, 1).sum((2, 3), keepdim=True) > 0
    gt_masks = None
    if training:
        gt_masks = (gt_alphas > 0).reshape(b, n_f, n_i, gt_alphas.shape[2], gt_alphas.shape[3])
        if gt_masks.shape[-1] != masks.shape[-1]:
            gt_masks = resize_any_shape(gt_masks, masks.shape[-1] * 1.0 / gt_masks.shape[-1], use_max_pool=True)
    x = os32_to_os8(sd, p, x, mid_fea, training)
    return x, masks, valid_masks, gt_masks


def decoder_image(sd, p, x, mid_fea, b, n_f, n_i, masks, it, gt_alphas, training, dcfg):
    """ResShortCut_InstMattSpconv_Dec.forward -- resnet_inst_matt_spconv.py:292-344."""
    fea1, fea2, fea3 = mid_fea['shortcut'][:3]
    image = mid_fea['image']
    h, w = image.shape[-2:]
    x, masks, valid_masks, gt_masks = _dec_prologue(sd, p, x, mid_fea, b, n_f, n_i, masks, gt_alphas, training)
    x_os8, x, queries, loss_max_atten, _ = imd(sd, p + '.refine_OS8', x, masks, training, gt_masks,
                                               n_block=dcfg.get('atten_block', 2), max_inst=dcfg.get('max_inst', 10))
    x_os8 = F.interpolate(x_os8, size=(h, w), mode='bilinear', align_corners=False)
    x_os8 = (torch.tanh(x_os8) + 1.0) / 2.0
    x_os8 = x_os8 * valid_masks if training else x_os8[:, :n_i]

    warm = dcfg.get('warmup_detail_iter', 3000)
    guided = x_os8.clone()
    use_gt = False
    if training and (it < warm or x_os8.sum() == 0 or (it < warm * 3 and random.random() < 0.5)):
        guided = gt_alphas.clone()
        use_gt = True
    unknown_os8 = compute_unknown(guided, 30)
    x_os4, x_os1 = process_os4_os1(sd, p, x, b, n_f, fea1, fea2, fea3, image, x_os8, queries, guided, unknown_os8, training)
    ret = {'alpha_os1': x_os1, 'alpha_os4': x_os4, 'alpha_os8': x_os8}
    alpha, w4, w1 = dec_fuse(ret, unknown_os8, training)
    ret['refined_masks'] = alpha
    if use_gt:
        w4 = compute_unknown(gt_alphas, 30, training) * unknown_os8
        w1 = compute_unknown(gt_alphas, 15, training) * unknown_os8
    ret['weight_os4'] = w4
    ret['weight_os1'] = w1
    ret['detail_mask'] = unknown_os8
    if training and it >= dcfg.get('warmup_mask_atten_iter', 4000):
        ret['loss_max_atten'] = loss_max_atten
    return ret


# ----------------------------------------------------------------------------------------------------
# temporal decoder (resnet_inst_matt_spconv_temp.py)
# ----------------------------------------------------------------------------------------------------

def gaussian_smoothing(x, sigma):
    """maggie/utils/utils.py:61-83, including its separable-kernel quirk (kernel = g*g broadcast over rows)."""
    ks = sigma * 2 + 1
    pad = ks // 2
    xp = F.pad(x, (pad, pad, pad, pad), mode='constant', value=0)
    grid = torch.arange(ks).float() - ks // 2
    g = torch.exp(-grid ** 2 / (2 * sigma ** 2))
    g = g / g.sum()
    k = (g.view(1, 1, -1) * g.view(1, 1, -1)).expand(x.shape[1], 1, ks, ks).type_as(x)
    sm = F.conv2d(xp, k, stride=1, padding=0, groups=x.shape[1])
    sm = sm[:, :, pad:-pad, pad:-pad]
    return F.interpolate(sm, size=x.shape[-2:], mode='bilinear', align_corners=False)


def diff_module(sd, p, x, training):
    """diff_module -- resnet_inst_matt_spconv_temp.py:25-33."""
    x = F.relu(bn(sd, p + '.1', sn_conv(sd, p + '.0', x), training))
    x = F.relu(bn(sd, p + '.4', sn_conv(sd, p + '.3', x, 1, 1), training))
    return F.conv2d(x, sd[p + '.6.weight'], None, 1, 1)


def bidirectional_fusion(sd, p, feat, preds, training):
    """resnet_inst_matt_spconv_temp.py:35-79."""
    n_f = feat.shape[1]
    fdiffs, bdiffs = [], []
    fpreds = [preds[:, 0]]
    bpreds = [preds[:, n_f - 1]]
    for i in range(1, n_f):
        d = diff_module(sd, p + '.diff_module', torch.cat([feat[:, i - 1], feat[:, i]], 1), training)
        d = F.interpolate(d, scale_factor=8.0, mode='bilinear', align_corners=False)
        fdiffs.append(d)
        fpreds.append(fpreds[-1] * (1 - d.sigmoid()) + preds[:, i] * d.sigmoid())
    fdiffs = torch.stack([torch.zeros_like(fdiffs[0])] + fdiffs, 1)
    for i in range(n_f - 1, 0, -1):
        d = diff_module(sd, p + '.diff_module', torch.cat([feat[:, i], feat[:, i - 1]], 1), training)
        d = F.interpolate(d, scale_factor=8.0, mode='bilinear', align_corners=False)
        bdiffs.append(d)
        bpreds.append(bpreds[-1] * (1 - d.sigmoid()) + preds[:, i - 1] * d.sigmoid())
    bpreds = bpreds[::-1]
    bdiffs = bdiffs[::-1]
    bdiffs = torch.stack(bdiffs + [torch.zeros_like(bdiffs[-1])], 1)
    fuse = []
    for i in range(n_f):
        if i == 0:
            fuse.append(fpreds[i])
        elif i == n_f - 1:
            fuse.append(bpreds[i])
        else:
            fuse.append((fpreds[i] + bpreds[i]) / 2)
    return fdiffs, bdiffs, torch.stack(fuse, 1)


def loss_dtssd(pred, gt, mask):
    """maggie/network/loss.py:7-16."""
    dadt = pred[:, 1:] - pred[:, :-1]
    dgdt = gt[:, 1:] - gt[:, :-1]
    diff = (dadt - dgdt) ** 2 * mask[:, 1:]
    return torch.sum(diff) / torch.sum(mask[:, 1:] + 1e-6)


def loss_temporal_sparsity(diff_forward, diff_backward, spar_gt):
    """resnet_inst_matt_spconv_temp.py:183-203."""
    loss = {}
    spar_gt = spar_gt.view(diff_forward.shape[0], -1, *spar_gt.shape[1:])
    bf = F.binary_cross_entropy_with_logits(diff_forward[:, 1:, 0], spar_gt[:, 1:, 0], reduction='mean')
    bb = F.binary_cross_entropy_with_logits(diff_backward[:, :-1, 0], spar_gt[:, 1:, 0], reduction='mean')
    loss['loss_temp_bce'] = bf + bb
    ones = torch.ones_like(spar_gt[:, 1:, 0:1])
    df = loss_dtssd(diff_forward[:, 1:].sigmoid(), spar_gt[:, 1:, 0:1], ones)
    db = loss_dtssd(diff_backward[:, :-1].sigmoid(), spar_gt[:, 1:, 0:1], ones)
    loss['loss_temp_dtssd'] = df + db
    loss['loss_temp'] = (loss['loss_temp_bce'] + df + db) * 0.25
    return loss


def video_eval_region(x_os8, n_i, h, w):
    """Region ops of the video decoder at inference, resnet_inst_matt_spconv_temp.py:115-142: snap the coarse alpha (>= 0.95 -> 1), the
    unknown band of the snapped alpha, then per plane the padded bounding box of (gaussian-smoothed alpha > 0.1) applied to both.
    -> (cropped coarse alpha, detail mask). A function of x_os8 alone, so a test can feed it the product's own coarse alpha."""
    x_os8 = torch.where(x_os8 >= 0.95, torch.ones_like(x_os8), x_os8)
    unknown_os8 = compute_unknown(x_os8, 30)
    smooth = gaussian_smoothing(x_os8, 3)
    x_os8 = x_os8.clone()
    for i in range(smooth.shape[0]):
        for j in range(n_i):
            coarse = smooth[i, j] > 0.1
            ys, xs = torch.nonzero(coarse, as_tuple=True)
            if len(ys) == 0:
                continue
            y0 = max(0, int(ys.min()) - 30)
            y1 = min(int(ys.max()) + 30, h)
            x0 = max(0, int(xs.min()) - 30)
            x1 = min(int(xs.max()) + 30, w)
            tm = torch.zeros_like(coarse)
            tm[y0:y1, x0:x1] = 1
            unknown_os8[i, j] = unknown_os8[i, j] * tm
            x_os8[i, j] = x_os8[i, j] * tm
    return x_os8, unknown_os8


def decoder_video(sd, p, x, mid_fea, b, n_f, n_i, masks, it, gt_alphas, training, dcfg, mem_feat=None, spar_gt=None):
    """ResShortCut_InstMattSpconv_BiTempSpar_Dec.forward -- resnet_inst_matt_spconv_temp.py:81-181."""
    temp_method_full = dcfg.get('temp_method', 'bi')
    temp_method = temp_method_full.split('_')[0]
    use_fusion = 'fusion' in temp_method_full
    use_temp = temp_method_full != 'none'
    fea1, fea2, fea3 = mid_fea['shortcut'][:3]
    image = mid_fea['image']
    x, masks, valid_masks, gt_masks = _dec_prologue(sd, p, x, mid_fea, b, n_f, n_i, masks, gt_alphas, training)

    def prop(feat):
        return conv_gru_propagate(sd, p + '.os8_temp_module', feat, n_f, mem_feat, temp_method)

    x_os8, x, queries, loss_max_atten, hidden = imd(sd, p + '.refine_OS8', x, masks, training, gt_masks, prop,
                                                    n_block=dcfg.get('atten_block', 2), max_inst=dcfg.get('max_inst', 10))
    feat_os8 = x.view(b, n_f, *x.shape[1:]).detach()
    x_os8 = F.interpolate(x_os8, scale_factor=8.0, mode='bilinear', align_corners=False)
    x_os8 = (torch.tanh(x_os8) + 1.0) / 2.0
    x_os8 = x_os8 * valid_masks if training else x_os8[:, :n_i]
    warm = dcfg.get('warmup_detail_iter', 3000)
    guided = x_os8
    use_gt = False
    if training and (it < warm or x_os8.sum() == 0 or (it < warm * 3 and random.random() < 0.5)):
        guided = gt_alphas.clone()
        use_gt = True
    if not training:
        x_os8, unknown_os8 = video_eval_region(x_os8, n_i, *image.shape[-2:])
        guided = x_os8
    else:
        unknown_os8 = compute_unknown(guided, 30)
    x_os4, x_os1 = process_os4_os1(sd, p, x, b, n_f, fea1, fea2, fea3, image, x_os8, queries, guided, unknown_os8, training)
    ret = {'alpha_os1': x_os1, 'alpha_os4': x_os4, 'alpha_os8': x_os8}
    alpha, w4, w1 = dec_fuse(ret, unknown_os8, training)
    ret['refined_masks'] = alpha
    ret['detail_mask'] = unknown_os8
    if use_temp:
        ret['mem_feat'] = hidden
    if use_gt:
        w4 = compute_unknown(gt_alphas, 30, training) * unknown_os8
        w1 = compute_unknown(gt_alphas, 15, training) * unknown_os8
    ret['weight_os4'] = w4
    ret['weight_os1'] = w1
    temp_alpha = alpha.view(b, n_f, *alpha.shape[1:])
    df, db, fused = bidirectional_fusion(sd, p, feat_os8, temp_alpha, training)
    if (not training and use_fusion) or training:
        ret['temp_alpha'] = fused
        ret['diff_forward'] = df.sigmoid()
        ret['diff_backward'] = db.sigmoid()
    if training:
        ret['loss_max_atten'] = loss_max_atten
        ret.update(loss_temporal_sparsity(df, db, spar_gt))
    return ret


# ----------------------------------------------------------------------------------------------------
# losses (maggie/network/loss.py, arch/maggie.py:237-368)
# ----------------------------------------------------------------------------------------------------

def regression_loss(logit, target, weight):
    """MaGGIe.regression_loss l1 branch -- arch/maggie.py:255-262."""
    loss = F.l1_loss(logit * weight, target * weight, reduction='none')
    return loss.sum() / (torch.sum(weight) + 1e-8)


def _gauss_kernel(channels):
    k = torch.tensor([[1., 4., 6., 4., 1], [4., 16., 24., 16., 4.], [6., 24., 36., 24., 6.],
                      [4., 16., 24., 16., 4.], [1., 4., 6., 4., 1.]]) / 256.
    return k.repeat(channels, 1, 1, 1)


def _conv_gauss(img, kernel):
    img = F.pad(img, (2, 2, 2, 2), mode='reflect')
    return F.conv2d(img, kernel, groups=img.shape[1])


def _lap_upsample(x):
    """loss.py:51-58: zero-stuff x2 then 4*gauss."""
    n, c, h, w = x.shape
    up = x.new_zeros((n, c, h * 2, w * 2))
    up[:, :, ::2, ::2] = x
    return _conv_gauss(up, 4 * _gauss_kernel(c))


def lap_loss(inp, target, weight, max_levels=3, channels=3):
    """LapLoss.forward -- loss.py:120-191, with its channels=3-on-1-channel-input broadcast quirk."""
    kernel = _gauss_kernel(channels)

    def pyramid(img):
        cur = img
        pyr = []
        for _ in range(max_levels):
            down = _conv_gauss(cur, kernel)[:, :, ::2, ::2]
            pyr.append(cur - _lap_upsample(down))
            cur = down
        return pyr
    pi, pt = pyramid(inp), pyramid(target)
    total = 0
    cur_w = weight
    for i in range(max_levels):
        total = total + (F.l1_loss(pi[i], pt[i], reduction='none') * cur_w).sum() / (cur_w.sum() + 1e-6)
        cur_w = cur_w[:, :, ::2, ::2]
    return total


def grad_loss(logit, label, mask, eps=1e-6):
    """GradientLoss.forward -- loss.py:67-118."""
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
    kx = kx / kx.abs().sum()
    ky = kx.t().contiguous()

    def sobel(t):
        n, c, h, w = t.shape
        tp = F.pad(t.reshape(n * c, 1, h, w), pad=[1, 1, 1, 1], mode='replicate')
        gx = F.conv2d(tp, kx[None, None])
        gy = F.conv2d(tp, ky[None, None])
        return torch.sqrt(gx * gx + gy * gy + eps).reshape(n, c, h, w)
    logit = logit * mask
    label = label * mask
    return torch.sum(F.l1_loss(sobel(logit), sobel(label), reduction='none')) / (mask.sum() + eps)


def compute_loss(mcfg, pred, weight_os4, weight_os1, alphas, trans_gt, alpha_shape):
    """MaGGIe.compute_loss -- arch/maggie.py:268-368."""
    a1, a4, a8 = pred['alpha_os1'], pred['alpha_os4'], pred['alpha_os8']
    num_masks = mcfg['encoder_args']['num_mask']
    ld = {}
    weight_os8 = torch.ones_like(a8)
    valid = alphas.sum((2, 3), keepdim=True) > 0
    weight_os8 = weight_os8 * valid
    if mcfg.get('loss_reweight_os8', True):
        ug = (alphas <= 254.0 / 255.0) & (alphas >= 1.0 / 255.0)
        up = (a8 <= 254.0 / 255.0) & (a8 >= 1.0 / 255.0)
        weight_os8 = (ug | up).type(weight_os8.dtype) + weight_os8
    n_i = alphas.shape[1]
    if num_masks - n_i > 0:
        padding = torch.zeros((alphas.shape[0], num_masks - n_i, *alphas.shape[-2:]))
        alphas = torch.cat([alphas, padding], 1)
        trans_gt = torch.cat([trans_gt, padding], 1)
    total = 0
    if mcfg['loss_alpha_w'] > 0:
        r1 = regression_loss(a1, alphas, weight_os1)
        r4 = regression_loss(a4, alphas, weight_os4)
        r8 = regression_loss(a8, alphas, weight_os8)
        ld['loss_rec_os1'], ld['loss_rec_os4'], ld['loss_rec_os8'] = r1, r4, r8
        ld['loss_rec'] = r1 * 2 + r4 + r8
        total = total + ld['loss_rec'] * mcfg['loss_alpha_w']
    if mcfg['loss_alpha_lap_w'] > 0:
        h, w = a8.shape[-2:]
        v = lambda t: t.reshape(-1, 1, h, w)
        l1_ = lap_loss(v(a1), v(alphas), v(weight_os1))
        l4_ = lap_loss(v(a4), v(alphas), v(weight_os4))
        l8_ = lap_loss(v(a8), v(alphas), v(weight_os8))
        ld['loss_lap_os1'], ld['loss_lap_os4'], ld['loss_lap_os8'] = l1_, l4_, l8_
        ld['loss_lap'] = l1_ * 2 + l4_ + l8_
        total = total + ld['loss_lap'] * mcfg['loss_alpha_lap_w']
    if mcfg['loss_alpha_grad_w'] > 0:
        g1 = grad_loss(a1, alphas, weight_os1)
        g4 = grad_loss(a4, alphas, weight_os4)
        g8 = grad_loss(a8, alphas, weight_os8)
        ld['loss_grad_os1'], ld['loss_grad_os4'], ld['loss_grad_os8'] = g1, g4, g8
        ld['loss_grad'] = g1 * 2 + g4 + g8
        total = total + ld['loss_grad'] * mcfg['loss_alpha_grad_w']
    if mcfg.get('loss_dtSSD_w', 0) > 0:
        rs = lambda t: t.reshape(*alpha_shape)
        d1 = loss_dtssd(rs(a1), rs(alphas), rs(weight_os1))
        d4 = loss_dtssd(rs(a4), rs(alphas), rs(weight_os4))
        d8 = loss_dtssd(rs(a8), rs(alphas), rs(weight_os8))
        ld['loss_dtSSD_os1'], ld['loss_dtSSD_os4'], ld['loss_dtSSD_os8'] = d1, d4, d8
        ld['loss_dtSSD'] = d1 * 2 + d4 + d8
        total = total + ld['loss_dtSSD'] * mcfg['loss_dtSSD_w']
    ld['total'] = total
    return ld


# ----------------------------------------------------------------------------------------------------
# arch (maggie/network/arch/maggie.py:63-235, maggie_temp.py)
# ----------------------------------------------------------------------------------------------------

def maggie_forward(sd, mcfg, batch, training, mem_feat=None, prev_pred=None):
    """MaGGIe.forward / MaGGIe_Temp.forward. `mcfg` is the `model` section of the yaml as a plain dict.
    Consumes the GLOBAL numpy / random RNGs in the reference's order."""
    is_temp = mcfg.get('arch', 'MaGGIe') == 'MaGGIe_Temp'
    num_masks = mcfg['encoder_args']['num_mask']
    x = batch['image']
    masks = batch['mask']
    alphas = batch.get('alpha', None)
    trans_gt = batch.get('transition', None)
    b, n_f, _, h, w = x.shape
    n_i = masks.shape[2]
    x = x.reshape(-1, 3, h, w)
    if masks.shape[-1] != w:
        masks = F.interpolate(masks.flatten(0, 1), size=(h, w), mode='nearest')
    else:
        masks = masks.reshape(-1, n_i, h, w)
    # prepare_input -- arch/maggie.py:200-235
    chosen_ids = None
    inp_masks = masks
    if num_masks - n_i > 0:
        if not training:
            inp_masks = torch.cat([masks, torch.zeros((b * n_f, num_masks - n_i, h, w))], 1)
        else:
            chosen_ids = np.random.choice(num_masks, n_i, replace=False)
            inp_masks = torch.zeros((b * n_f, num_masks, h, w))
            inp_masks[:, chosen_ids] = masks
            masks = inp_masks
            if alphas is not None:
                na = torch.zeros((b, n_f, num_masks, h, w))
                na[:, :, chosen_ids] = alphas
                alphas = na
            if trans_gt is not None:
                nt = torch.zeros((b, n_f, num_masks, h, w))
                nt[:, :, chosen_ids] = trans_gt
                trans_gt = nt
            n_i = num_masks
    inp = torch.cat([x, inp_masks], 1)
    if alphas is not None:
        alphas = alphas.reshape(-1, n_i, h, w)
    if trans_gt is not None:
        trans_gt = trans_gt.reshape(-1, n_i, h, w)

    embedding, mid_fea = encoder(sd, 'encoder', inp, training)
    embedding = aspp(sd, 'aspp', embedding, training)
    dcfg = dict(mcfg['decoder_args'])
    it = batch.get('iter', 0)
    if is_temp:
        pred = decoder_video(sd, 'decoder', embedding, mid_fea, b, n_f, n_i, masks, it, alphas, training, dcfg,
                             mem_feat=mem_feat, spar_gt=trans_gt)
    else:
        pred = decoder_image(sd, 'decoder', embedding, mid_fea, b, n_f, n_i, masks, it, alphas, training, dcfg)

    alpha_pred = pred.pop('refined_masks')
    weight_os4 = pred['detail_mask'].type(alpha_pred.dtype)
    weight_os1 = weight_os4
    if training and 'weight_os4' in pred and np.random.rand() < 0.75:
        weight_os4 = pred.pop('weight_os4')
        weight_os1 = pred.pop('weight_os1')

    n_out = num_masks if (training and num_masks > 0) else n_i
    output = {}
    for k in ('alpha_os1', 'alpha_os4', 'alpha_os8', 'detail_mask'):
        output[k] = pred[k][:, :n_out].reshape(b, n_f, n_out, h, w)
    output['refined_masks'] = alpha_pred[:, :n_out].reshape(b, n_f, n_out, h, w)
    if is_temp:
        dbk = pred.pop('diff_backward', None)
        dfw = pred.pop('diff_forward', None)
        ta = pred.pop('temp_alpha', None)
        if dbk is not None:
            output['diff_pred_backward'] = dbk.repeat(1, 1, n_i, 1, 1)
            output['diff_pred_forward'] = dfw.repeat(1, 1, n_i, 1, 1)
            output['temp_alpha'] = ta

    if training:
        alphas = alphas.reshape(-1, n_i, h, w)
        trans_gt = trans_gt.reshape(-1, n_i, h, w)
        valid_masks = (trans_gt.sum((2, 3), keepdim=True) > 0).float()
        for k, v in list(pred.items()):
            if 'loss' in k or 'mem_' in k:
                continue
            pred[k] = v * valid_masks
        loss_dict = compute_loss(mcfg, pred, weight_os4, weight_os1, alphas, trans_gt, (b, n_f, num_masks, h, w))
        if 'loss_max_atten' in pred and mcfg['loss_atten_w'] > 0:
            loss_dict['loss_max_atten'] = pred['loss_max_atten']
            loss_dict['total'] = loss_dict['total'] + loss_dict['loss_max_atten'] * mcfg['loss_atten_w']
        if is_temp and 'loss_temp' in pred:
            loss_dict['loss_temp_bce'] = pred['loss_temp_bce']
            loss_dict['loss_temp'] = pred['loss_temp']
            loss_dict['total'] = loss_dict['total'] + pred['loss_temp']
            loss_dict['loss_temp_dtssd'] = pred['loss_temp_dtssd']
        if chosen_ids is not None:
            for k, v in output.items():
                output[k] = v[:, :, chosen_ids]
        return output, loss_dict

    for k, v in output.items():
        output[k] = v[:, :, :n_i]
    for k in pred:
        if k.startswith('mem_'):
            output[k] = pred[k]

    if is_temp:
        # MaGGIe_Temp.forward eval post-fusion -- arch/maggie_temp.py:34-77 (hard-wired to frames 0,1,2)
        alphas_o = output['refined_masks']
        pp = alphas_o[:, 0] if prev_pred is None else prev_pred
        nxt = alphas_o[:, -1]
        dfw = (output['diff_pred_forward'] > 0.5).float()
        dbk = (output['diff_pred_backward'] > 0.5).float()
        p01 = pp * (1 - dfw[:, 1]) + alphas_o[:, 1] * dfw[:, 1]
        p21 = nxt * (1 - dbk[:, 1]) + alphas_o[:, 1] * dbk[:, 1]
        diff = torch.abs(p01 - p21)
        p01 = torch.where(diff > 0.0, alphas_o[:, 1], p01)
        alphas_o[:, 1] = p01
        p12 = p01 * (1 - dfw[:, 2]) + nxt * dfw[:, 2]
        alphas_o[:, 2] = p12
    return output
