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

from ... import functional as MF


def _need_hip_attention(n_tokens, d):
    """The HIP attention kernels are built for the configuration of maggie_{image,video}.yaml (10 instance tokens, width 128). Anything
    else is rejected: there is no torch fallback anywhere in this build (maggie_amd/hip.py)."""
    if n_tokens != 10 or d != 128:
        raise MF.K.hip.MaggieHipError('MaGGIe (MI355X build): the cross-attention kernels are built for max_inst = 10 tokens of width 128 '
                                      '(configs/maggie_{image,video}.yaml); got %d tokens of width %d' % (n_tokens, d))


def _split_in_proj(mha):
    # three views whose gradients are written straight into the slices of ONE (3d, d) / (3d) buffer by the kernels that produce them
    # (functional.split_packed; was: unbind, whose backward is a stack of three tensors per packed parameter and step)
    return MF.split_packed(mha.in_proj_weight, 3), MF.split_packed(mha.in_proj_bias, 3)


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
        """tgt: (b, T, d) tokens; key padding mask (b, T) True = ignore. Five HIP launches: three projections (position added inside),
        the T x T attention core, out-projection + residual + LayerNorm (mg_token_linear_*, mg_token_sa_*)."""
        (wq, wk, wv), (bq, bk, bv) = _split_in_proj(self.self_attn)
        tg = MF.Fan(tgt, 4)                                        # four consumers: their gradients meet in one launch (functional.FanOut)
        q, k, v = MF.token_linear_multi([dict(x=tg(), W=wq, b=bq, xadd=MF.take(query_pos)), dict(x=tg(), W=wk, b=bk, xadd=MF.take(query_pos)),
                                         dict(x=tg(), W=wv, b=bv)])                                      # three projections, one launch
        ctx = MF.token_self_attention(q, k, v, tgt_key_padding_mask)
        return MF.token_linear(ctx, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, res=tg(), ln=self.norm)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert nhead == 1 and dropout == 0.0 and not normalize_before
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    # Each direction is: two levels of small token-side linears (level 2 reads level 1), the pass over the feature rows, the rest. The levels are
    # exposed as layer lists so that the caller can batch levels of DIFFERENT blocks that read the same tokens into one launch
    # (InstanceMatteDecoder: features_from_tokens of block i and tokens_from_features of block i + 1 both start from the tokens block i's
    # self-attention produced).

    def tff_level1(self, tokens, token_pos, id_table):
        (wq, wk, wv), (bq, bk, bv) = _split_in_proj(self.multihead_attn)      # ONE split per forward: the later stages reuse the slices
        if id_table is not None:
            wk = MF.Fan(wk, 2)                                    # two consumers (here and level 2): their gradients meet in the packed buffer's slice
        self._proj = (wq, wk, wv), (bq, bk, bv)
        # (tokens / token_pos / id_table: tensors, or functional.Fan objects handing out one alias per consumer)
        layers = [dict(x=MF.take(tokens), W=wq, b=bq, xadd=MF.take(token_pos))]              # q
        if id_table is not None:
            layers.append(dict(x=MF.take(id_table), W=MF.take(wk), b=bk))                    # key_pos (n_id, d): E[id] Wk^T + bk
        return layers

    def tff_level2(self, res1, id_table):
        (wq, wk, wv), (bq, bk, bv) = self._proj
        q = res1[0]
        key_pos = res1[1] if id_table is not None else bk[None, :]
        # q Wk (fold Wk into the queries; wk used as (K, N): no transposed copy) and the (b,T,n_id) score-bias table q . key_pos
        return [dict(x=q, W=MF.take(wk), wt=True), dict(x=q, W=key_pos)]

    def tff_finish(self, tokens, res2, feat, feat_ids):
        (wq, wk, wv), (bq, bk, bv) = self._proj
        qk, tbl = res2
        tokens = MF.take(tokens)
        d = tokens.shape[-1]
        _need_hip_attention(tokens.shape[1], d)
        p, ctx = MF.attn_tokens_from_features(qk, tbl, MF.take(feat), feat_ids, 1.0 / math.sqrt(d))
        self._proj = None
        h = MF.token_linear(ctx, wv, bv)
        return MF.token_linear(h, self.multihead_attn.out_proj.weight, self.multihead_attn.out_proj.bias, res=tokens, ln=self.norm), p

    def tokens_from_features(self, tokens, token_pos, feat, feat_ids, id_table, pre=None):
        """tokens (b,T,d) <- feat (b,L,d) with key position = id_table[feat_ids] ((b,L) int32). Returns new tokens and the
        attention matrix (b,T,L). The pass over the feature rows (scores, softmax over L, context) is one HIP pipeline
        (mg_attn_tok_fwd / _bwd); the 10-token projections around it are fused HIP linears (mg_token_linear_*). `pre`: the level-2 results
        (qk, tbl) when the caller already computed them in a batched launch."""
        if pre is None:
            res1 = MF.token_linear_multi(self.tff_level1(tokens, token_pos, id_table))
            pre = MF.token_linear_multi(self.tff_level2(res1, id_table))
        return self.tff_finish(tokens, pre, feat, feat_ids)

    def fft_level1(self, tokens, token_pos, id_table):
        (wq, wk, wv), (bq, bk, bv) = _split_in_proj(self.multihead_attn)
        if id_table is not None:
            wq = MF.Fan(wq, 2)                                    # two consumers (here and level 2)
        self._proj = (wq, wk, wv), (bq, bk, bv)
        layers = [dict(x=MF.take(tokens), W=wk, b=bk, xadd=MF.take(token_pos)), dict(x=MF.take(tokens), W=wv, b=bv)]    # k (b,T,d), v
        if id_table is not None:
            layers.append(dict(x=MF.take(id_table), W=MF.take(wq), b=bq))                    # qry_pos (n_id, d)
        return layers

    def fft_level2(self, res1, id_table):
        (wq, wk, wv), (bq, bk, bv) = self._proj
        self._proj = None
        k, v = res1[0], res1[1]
        qry_pos = res1[2] if id_table is not None else bq[None, :]
        # vp (b,T,d): rows of (Wo V^T)^T; kq (b,T,d): Wq folded into the keys; the (b,T,n_id) score-bias table
        return [dict(x=v, W=self.multihead_attn.out_proj.weight), dict(x=k, W=MF.take(wq), wt=True), dict(x=k, W=qry_pos)]

    def fft_finish(self, feat, feat_ids, res2, token_padding_mask, n_tokens):
        vp, kq, tbl = res2
        feat, res = MF.take(feat), MF.take(feat)                 # (a functional.Fan hands out one alias per consumer: the attention and the residual)
        d = feat.shape[-1]
        _need_hip_attention(n_tokens, d)
        # tbl stays (b,T,n_id) as the token-side linear wrote it: the kernels index it that way (tn) -- was a transposed copy each way per block
        out = MF.attn_features_from_tokens(feat, kq, tbl, vp, self.multihead_attn.out_proj.bias, token_padding_mask, feat_ids,
                                           1.0 / math.sqrt(d), tn=True)
        return MF.rows_add_layernorm(res, out, self.norm)

    def features_from_tokens(self, feat, feat_ids, id_table, tokens, token_pos, token_padding_mask, pre=None):
        """feat (b,L,d) <- tokens (b,T,d); query position = id_table[feat_ids]. Scores, masked softmax over the T tokens and
        the value mix are one HIP kernel per direction (mg_attn_feat_fwd / _bwd); residual + LayerNorm over the b * L rows one more.
        `pre`: the level-2 results (vp, kq, tbl) when the caller already computed them in a batched launch."""
        n_tokens = (tokens.t if isinstance(tokens, MF.Fan) else tokens).shape[1]
        if pre is None:
            res1 = MF.token_linear_multi(self.fft_level1(tokens, token_pos, id_table))
            pre = MF.token_linear_multi(self.fft_level2(res1, id_table))
        return self.fft_finish(feat, feat_ids, pre, token_padding_mask, n_tokens)


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
        """Token rows (b, T, d): two fused HIP launches (linear + ReLU; linear + residual + LayerNorm). The instance-specific FFN of
        the sparse head (dropout 0.1, row counts in a device word) runs inside maggie_amd.sparse_head instead."""
        if self.dropout.p > 0 and self.training:
            raise MF.K.hip.MaggieHipError('FFNLayer with dropout on token rows is not part of maggie_{image,video}.yaml (see sparse_head)')
        h = MF.token_linear(tgt, self.linear1.weight, self.linear1.bias, relu=True)
        return MF.token_linear(h, self.linear2.weight, self.linear2.bias, res=tgt, ln=self.norm)


class MLP(nn.Module):
    """Very simple multi-layer perceptron (also called FFN)"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x, ln=None):
        """`ln`: a LayerNorm fused behind the last layer (InstanceMatteDecoder: decoder_norm(final_mlp(tokens)))."""
        if not x.is_cuda:
            raise MF.K.hip.MaggieHipError('MaGGIe (MI355X build) runs on the GPU only (MLP got a %s tensor)' % x.device.type)
        # token rows of any count (mg_token_linear_* tile over rows): per-process batches beyond 12 stay on the HIP kernels too
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            x = MF.token_linear(x, layer.weight, layer.bias, relu=not last, ln=ln if last else None)
        return x
