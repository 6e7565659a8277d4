"""InstanceMatteDecoder -- mirrors maggie/network/module/instance_matte_decoder.py:9-307 at the target configuration
(atten_stride=1, 1 head, use_id_pe=True, use_temp_pe=False, use_mask_atten=False).

Inputs/outputs are NHWC: ori_feat (b*n_f, h, w, C); mask (b, n_f, n_i, H, W) float. Returns
(logits (b*n_f, h, w, 16 [10 real]), out_feat (b*n_f, h, w, 64), tokens (b, 10, 64), max_loss, hidden_state)."""
import torch
import torch.nn as nn
from torch.nn import functional as F

from ... import functional as MF
from .base import ConvWeight, Marker
from .mask_attention import MLP, SelfAttentionLayer, CrossAttentionLayer, FFNLayer


def check_tokens(tokens):
    """mask_attention.py:95-98: NaN queries mean an empty guidance mask poisoned the attention."""
    if bool(torch.isnan(tokens).any()):
        raise ValueError("Mask is empty")


class InstanceMatteDecoder(nn.Module):
    def __init__(self, input_dim=256, atten_stride=1.0, attention_dim=256, n_block=2, n_head=4, output_dim=32, return_feat=True,
                 max_inst=10, use_temp_pe=True, use_id_pe=True):
        super().__init__()
        assert atten_stride == 1 and not use_temp_pe, 'only the configuration of maggie_{image,video}.yaml is built'
        self.n_block = n_block
        self.atten_dim = attention_dim
        self.atten_stride = atten_stride
        self.return_feat = return_feat
        self.max_inst = max_inst
        self.use_id_pe = use_id_pe
        self.feat_proj = MLP(input_dim, attention_dim, attention_dim, 1)
        self.sa_layers = nn.ModuleList()
        self.token_feat_ca_layers = nn.ModuleList()
        self.mlp_layers = nn.ModuleList()
        self.feat_token_ca_layers = nn.ModuleList()
        for _ in range(n_block):
            self.sa_layers.append(SelfAttentionLayer(attention_dim, n_head))
            self.token_feat_ca_layers.append(CrossAttentionLayer(attention_dim, n_head))
            self.mlp_layers.append(FFNLayer(attention_dim, attention_dim, 0.0))
            self.feat_token_ca_layers.append(CrossAttentionLayer(attention_dim, n_head))
        self.final_token_feat_ca = CrossAttentionLayer(attention_dim, n_head)
        self.final_mlp = MLP(attention_dim, attention_dim, output_dim, 1)
        self.decoder_norm = nn.LayerNorm(output_dim)
        self.n_temp_embed = 0
        self.n_id_embed = self.atten_dim
        self.query_feat = nn.Embedding(max_inst, attention_dim)
        self.id_embedding = nn.Embedding(max_inst + 1, self.n_id_embed)
        nn.init.xavier_uniform_(self.id_embedding.weight)
        nn.init.xavier_uniform_(self.query_feat.weight)
        self.conv = nn.Sequential(
            ConvWeight(attention_dim, attention_dim, 3, 1, 1, 1), nn.BatchNorm2d(attention_dim), Marker('LeakyReLU(0.2)'),
            ConvWeight(attention_dim, output_dim, 1, 1, 0, 1), nn.BatchNorm2d(output_dim), Marker('LeakyReLU(0.2)'))
        for m in self.conv:
            if isinstance(m, ConvWeight):
                nn.init.xavier_uniform_(m.weight)

    def compute_atten_loss(self, b, n_f, guidance_mask, atten_mat):
        """((guidance.sum(2) != 0) - (guidance * atten).sum(2)).sum() / (n_f * b) as one fused HIP reduction (mg_atten_loss_fwd / _bwd)."""
        return MF.atten_guidance_loss(guidance_mask, atten_mat, 1.0 / (n_f * b))

    def _smooth(self, x):
        c0, bn0, _, c1, bn1, _ = self.conv
        x = MF.conv_bn_act(x, MF.plain_krsc(c0, x.dtype, keep=True), bn0, MF.ACT_LRELU, 3, 3, 1, 1, 1, link_out=True)    # keep: applied twice per video forward
        return MF.conv_bn_act(x, MF.plain_krsc(c1, x.dtype, keep=True), bn1, MF.ACT_LRELU, 1, 1, 1, 0, 1)

    def plain_convs(self):
        return [self.conv[0], self.conv[3]]

    def forward(self, ori_feat, mask, use_mask_atten=False, gt_mask=None, aggregate_mem_fn=None):
        assert not use_mask_atten
        N, h, w, C = ori_feat.shape
        b, n_f, n_in = mask.shape[:3]
        dt = ori_feat.dtype
        # mask -> OS8 binary (resizeAnyShape(..., use_avg_pool_binary=True), utils.py:16-21), the ID position of every feature pixel = max over
        # instances of id * mask (:150-153), the token validity and (training) the ground-truth guidance: ONE launch (mg_imd_prep)
        n_i = self.max_inst
        stride = mask.shape[-1] // w
        if stride < 1 or mask.shape[-1] != w * stride or mask.shape[-2] != h * stride:
            raise MF.K.hip.MaggieHipError('InstanceMatteDecoder: the guidance mask must be an integer multiple of the OS8 map (%dx%d), got %dx%d'
                                          % (h, w, mask.shape[-2], mask.shape[-1]))
        mk = mask.float().contiguous()
        gm = gs = None
        if self.training:
            gm = gt_mask.float().contiguous()
            gs = gm.shape[-1] // w
            if gs < 1 or gm.shape[-1] != w * gs or gm.shape[-2] != h * gs:
                raise MF.K.hip.MaggieHipError('InstanceMatteDecoder: the ground-truth alphas must be an integer multiple of the OS8 map')
        L = n_f * h * w
        feat_ids = torch.empty((b, L), dtype=torch.int32, device=mask.device)
        valid_u8 = MF.ARENA.acc((b * n_i + 3) // 4, mask.device).view(torch.uint8)               # zeroed by the callee (or the graph's zero arena)
        guidance_mask = torch.empty((b, n_i, L), dtype=torch.float32, device=mask.device) if self.training else None
        MF.K.hip.call('mg_imd_prep', MF.K.hip.ptr(mk), MF.K.c_int(n_in), MF.K.c_int(stride), MF.K.hip.ptr(gm), MF.K.c_int(0 if gm is None else gm.shape[2]),
                      MF.K.c_int(gs or 1), MF.K.c_int(b), MF.K.c_int(n_f), MF.K.c_int(h), MF.K.c_int(w), MF.K.c_int(n_i), MF.K.hip.ptr(feat_ids),
                      MF.K.hip.ptr(guidance_mask), MF.K.hip.ptr(valid_u8), MF.K.hip.stream())
        token_padding_mask = valid_u8[:b * n_i].view(b, n_i) == 0
        id_table = self.id_embedding.weight.float() if self.use_id_pe else None
        # materialised once: every token-side launch below wants dense (b, 10, d) operands (an expanded view would be copied per use)
        token_pos = self.id_embedding.weight[1:self.max_inst + 1].float()[None].expand(b, -1, -1).contiguous()
        tokens = self.query_feat.weight.float()[None].expand(b, -1, -1).contiguous()

        # feature projection (Linear 128->128 over all rows) on the implicit-GEMM kernel
        lin = self.feat_proj.layers[0]
        wproj = MF._pad_krsc(lin.weight[:, None, :], dt, None, None)
        feat = MF.linear_rows(ori_feat.reshape(-1, C), wproj, lin.bias.float()).float().view(b, n_f * h * w, -1)

        max_loss, atten_terms = 0, []
        # the position embedding (13 consumers), the ID table (7) and every block's tokens (4) are handed out as one alias per consumer
        # (functional.Fan): the gradients of a tensor then meet in ONE launch instead of consumer-count - 1 autograd add kernels (~30 per step)
        pos = MF.Fan(token_pos, 16)
        idt = MF.Fan(id_table, 10) if id_table is not None else None
        pos_t = pos if self.use_id_pe else None
        tbl = idt if self.use_id_pe else None
        if not self.use_id_pe:
            feat_ids = torch.zeros_like(feat_ids)

        # The attention blocks run in fp32 with autocast OFF: every operand is a small fp32 tensor (10 tokens per sample, one
        # (L x 128) feature matrix), so autocast would only add a cast kernel per operand per op (~250 launches per step)
        with torch.autocast('cuda', enabled=False):
            pre = None                                            # (qk, tbl) of the NEXT tokens <- features block, computed one block early
            for i in range(self.n_block):
                # this version of the feature rows has three consumers (tokens <- features here, then the features <- tokens attention and its
                # residual): one alias each, their 8 MB gradients meet in one launch instead of two autograd adds
                feat = MF.Fan(feat, 3)
                tokens, att = self.token_feat_ca_layers[i].tokens_from_features(tokens, pos_t, feat, feat_ids, tbl, pre=pre)
                if self.training:
                    atten_terms.append(self.compute_atten_loss(b, n_f, guidance_mask, att))
                tokens = self.mlp_layers[i](tokens)
                tokens = self.sa_layers[i](tokens, tgt_key_padding_mask=token_padding_mask, query_pos=pos)
                # both cross attentions that follow start from THESE tokens (the features <- tokens block does not change them): their two
                # levels of token-side linears run as ONE launch per level (5 independent layers each) instead of two
                fft = self.feat_token_ca_layers[i]
                tk = MF.Fan(tokens, 4)                            # k and v of the features <- tokens block, q of the next block, its residual
                if not MF.TOKEN_XBLOCK:
                    feat = fft.features_from_tokens(feat, feat_ids, tbl, tk, pos_t, token_padding_mask)
                    tokens = tk
                    continue
                last = i + 1 == self.n_block
                nxt = self.final_token_feat_ca if last else self.token_feat_ca_layers[i + 1]
                nxt_pos, nxt_tbl = (pos, idt) if last else (pos_t, tbl)
                l1a, l1b = fft.fft_level1(tk, pos_t, tbl), nxt.tff_level1(tk, nxt_pos, nxt_tbl)
                r1 = MF.token_linear_multi(l1a + l1b)
                l2a, l2b = fft.fft_level2(r1[:len(l1a)], tbl), nxt.tff_level2(r1[len(l1a):], nxt_tbl)
                r2 = MF.token_linear_multi(l2a + l2b)
                feat = fft.features_from_tokens(feat, feat_ids, tbl, tokens, pos_t, token_padding_mask, pre=r2[:len(l2a)])
                pre = r2[len(l2a):]
                tokens = tk                                       # the next tokens <- features block takes its residual alias (tff_finish)
            tokens, att = self.final_token_feat_ca.tokens_from_features(tokens, pos, feat, feat_ids, idt, pre=pre)
            if self.training:
                atten_terms.append(self.compute_atten_loss(b, n_f, guidance_mask, att))
            if atten_terms:                                       # mean over the n_block + 1 attention maps: one launch (mg_scalar_lincomb)
                max_loss = MF.scalar_lincomb(atten_terms, [1.0 / (self.n_block + 1)] * len(atten_terms))
        if MF.EAGER_TOKEN_CHECK and not torch.cuda.is_current_stream_capturing():
            check_tokens(tokens)                                  # direct module use; MaGGIe.forward reads all its flags in ONE device->host copy

        feat = feat.to(dt).view(N, h, w, -1)
        hidden_state = None
        if aggregate_mem_fn is not None:
            no_temp_feat = feat
            feat, hidden_state = aggregate_mem_fn(feat.view(b, n_f, h, w, -1))
            feat = feat.flatten(0, 1)
            out_feat = self._smooth(no_temp_feat)
            feat = self._smooth(feat)
        else:
            feat = self._smooth(feat)
            out_feat = feat

        with torch.autocast('cuda', enabled=False):
            tokens = self.final_mlp(tokens, ln=self.decoder_norm)                               # (b, 10, c_out) fp32
        # einsum('bqc,btchw->btqhw'): a per-batch-element 1x1 conv whose weights are the tokens (padded to 16 outputs)
        cq = 16
        if n_i > cq or feat.shape[-1] not in (32, 64):
            raise MF.K.hip.MaggieHipError('InstanceMatteDecoder (MI355X build): the token einsum kernel is built for max_inst <= 16 and output_dim 32 / 64 '
                                          '(configs/maggie_{image,video}.yaml); got %d tokens of width %d' % (n_i, feat.shape[-1]))
        output_mask = MF.token_einsum(feat.view(b, n_f * h * w, -1), tokens).view(N, h, w, cq)      # one launch each way (mg_token_einsum_*)
        return output_mask, out_feat, tokens, max_loss, hidden_state
