"""Attention blocks of the instance matte decoder -- mirror maggie/network/module/mask_attention.py:9-206
(SelfAttentionLayer / CrossAttentionLayer / FFNLayer / MLP, post-norm, dropout 0, ONE head).

MI355X-first restructuring: one side of every cross attention is only `max_inst` (=10) tokens wide, so the big-side
Q/K/V projections (L x 128 x 128 GEMMs, L = n_f*h*w) are never materialised. The token side is folded into the
projection weights and the ID position embedding into an (max_inst+1)-row lookup table:

  tokens <- features :  score[q, l] = (Q_q Wk) . f_l + (Q_q . (E[id_l] Wk^T + bk)),   ctx_q = sum_l P[q,l] f_l,
                        out_q = (ctx_q Wv^T + bv) Wo^T + bo                      (sum_l P = 1)
  features <- tokens :  score[l, j] = f_l . (Wq^T K_j) + ((E[id_l] Wq^T + bq) . K_j),  out_l = sum_j P[l,j] (V_j Wo^T) + bo

which is algebraically identical to nn.MultiheadAttention(q + pos_q, k + pos_k, v) and turns each block into one
pass over the (L x 128) feature rows -- done by the HIP kernels of maggie_amd/csrc/attention.hip (forward and exact
backward). The 10-token projections around them are small fp32 torch ops.

The reference raises ValueError("Mask is empty") when a query/feature tensor holds NaN (mask_attention.py:95-98,129-132).
NaNs propagate through every later block, so the check is made ONCE on the final tokens (InstanceMatteDecoder.forward; one
host sync instead of five, and none inside a captured hipGraph).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ... import functional as MF


def _hip_attention(n_tokens, d):
    """The HIP attention kernels are built for the configuration of maggie_{image,video}.yaml (10 instance tokens, width 128)."""
    return n_tokens == 10 and d == 128


def _split_in_proj(mha):
    d = mha.embed_dim
    # unbind, not three slices: its backward is ONE stack of the three gradients (a slice costs zeros + copy + add each)
    return mha.in_proj_weight.view(3, d, d).unbind(0), mha.in_proj_bias.view(3, d).unbind(0)


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert nhead == 1 and dropout == 0.0 and not normalize_before
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, tgt_key_padding_mask=None, query_pos=None):
        """tgt: (b, T, d) tokens; key padding mask (b, T) True = ignore."""
        (wq, wk, wv), (bq, bk, bv) = _split_in_proj(self.self_attn)
        qk = tgt if query_pos is None else tgt + query_pos
        q = F.linear(qk, wq, bq)
        k = F.linear(qk, wk, bk)
        v = F.linear(tgt, wv, bv)
        s = torch.matmul(q, k.transpose(1, 2)) / math.sqrt(q.shape[-1])
        if tgt_key_padding_mask is not None:
            s = s.masked_fill(tgt_key_padding_mask[:, None, :], float('-inf'))
        p = torch.softmax(s, -1)
        out = self.self_attn.out_proj(torch.matmul(p, v))
        return self.norm(tgt + out)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert nhead == 1 and dropout == 0.0 and not normalize_before
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def tokens_from_features(self, tokens, token_pos, feat, feat_ids, id_table):
        """tokens (b,T,d) <- feat (b,L,d) with key position = id_table[feat_ids] ((b,L) int32). Returns new tokens and the
        attention matrix (b,T,L). The pass over the feature rows (scores, softmax over L, context) is one HIP pipeline
        (mg_attn_tok_fwd / _bwd); the 10-token projections around it are small fp32 torch ops."""
        (wq, wk, wv), (bq, bk, bv) = _split_in_proj(self.multihead_attn)
        d = tokens.shape[-1]
        q = F.linear(tokens if token_pos is None else tokens + token_pos, wq, bq)            # (b,T,d)
        qk = torch.matmul(q, wk)                                                             # fold Wk into the queries
        qb = (q * bk).sum(-1, keepdim=True)
        if id_table is not None:
            tbl = torch.matmul(q, (F.linear(id_table, wk)).t()) + qb                         # (b,T,n_id)
        else:
            tbl = qb                                                                         # (b,T,1); feat_ids are all 0
        if _hip_attention(tokens.shape[1], d):
            p, ctx = MF.attn_tokens_from_features(qk, tbl, feat, feat_ids, 1.0 / math.sqrt(d))
        else:                                                                                # other widths: plain torch
            s = torch.matmul(qk, feat.transpose(1, 2)) + torch.gather(tbl, 2, feat_ids.long()[:, None, :].expand(-1, q.shape[1], -1))
            p = torch.softmax(s / math.sqrt(d), -1)
            ctx = torch.matmul(p, feat)                                                      # (b,T,d)
        out = self.multihead_attn.out_proj(F.linear(ctx, wv, bv))
        return self.norm(tokens + out), p

    def features_from_tokens(self, feat, feat_ids, id_table, tokens, token_pos, token_padding_mask):
        """feat (b,L,d) <- tokens (b,T,d); query position = id_table[feat_ids]. Scores, masked softmax over the T tokens and
        the value mix are one HIP kernel per direction (mg_attn_feat_fwd / _bwd)."""
        (wq, wk, wv), (bq, bk, bv) = _split_in_proj(self.multihead_attn)
        d = feat.shape[-1]
        k = F.linear(tokens if token_pos is None else tokens + token_pos, wk, bk)            # (b,T,d)
        vp = F.linear(F.linear(tokens, wv, bv), self.multihead_attn.out_proj.weight)         # (b,T,d): rows of (Wo V^T)^T
        kq = torch.matmul(k, wq)                                                             # (b,T,d): fold Wq into the keys
        if id_table is not None:
            tbl = torch.matmul(F.linear(id_table, wq, bq), k.transpose(1, 2))                # (b,n_id,T)
        else:
            tbl = torch.matmul(k, bq)[:, None, :]                                            # (b,1,T)
        if _hip_attention(tokens.shape[1], d):
            out = MF.attn_features_from_tokens(feat, kq, tbl, vp, self.multihead_attn.out_proj.bias, token_padding_mask, feat_ids,
                                               1.0 / math.sqrt(d))
        else:
            s = torch.matmul(feat, kq.transpose(1, 2)) + torch.gather(tbl, 1, feat_ids.long()[:, :, None].expand(-1, -1, k.shape[1]))
            s = s / math.sqrt(d)
            if token_padding_mask is not None:
                s = s.masked_fill(token_padding_mask[:, None, :], float('-inf'))
            out = torch.matmul(torch.softmax(s, -1), vp) + self.multihead_attn.out_proj.bias
        return self.norm(feat + out)


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt):
        tgt2 = self.linear2(self.dropout(F.relu(self.linear1(tgt))))
        return self.norm(tgt + self.dropout(tgt2))


class MLP(nn.Module):
    """Very simple multi-layer perceptron (also called FFN)"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return x
