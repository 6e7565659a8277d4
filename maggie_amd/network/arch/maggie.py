"""MaGGIe -- mirrors maggie/network/arch/maggie.py:18-368: same constructor (`CfgNode | dict`), same
`forward(batch, **kwargs)` contract (eval -> dict of (b, n_f, n_i, h, w) tensors; train -> (dict, loss_dict)),
same loss names, same host-RNG consumption order (np.random / random), same state_dict keys."""
import logging

import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

try:                                                     # optional: the reference mixes this in for from_pretrained()
    from huggingface_hub import PyTorchModelHubMixin
except Exception:                                        # pragma: no cover
    class PyTorchModelHubMixin:                          # type: ignore
        pass

from ...utils.config import CfgNode
from ... import functional as MF
from ... import kernels as K
from ..module import ASPP
from ..encoder import *      # noqa: F401,F403  (factory names are resolved by eval(), like the reference)
from ..decoder import *      # noqa: F401,F403


def _bn_counted(method):
    """Stage methods of the VIDEO model (its BatchNorm layers are called a varying number of times per forward, so the image model's
    one-foreach-per-step counter bump does not apply): BatchNorm calls inside the stage are logged and the counters bumped at its end with
    one foreach launch per multiplicity -- inside the stage, so it is captured with it."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *a, **k):
        if not self.training or MF.DEFER_BN_COUNTERS or MF.BN_COUNT_LOG is not None:
            return method(self, *a, **k)
        MF.BN_COUNT_LOG = log = []
        try:
            out = method(self, *a, **k)
        finally:
            MF.BN_COUNT_LOG = None
        MF.flush_bn_count_log(log)
        return out
    return wrapper


class MaGGIe(nn.Module, PyTorchModelHubMixin):
    def __init__(self, cfg):
        super().__init__()
        if isinstance(cfg, dict) and not hasattr(cfg, 'encoder_args'):
            cfg = CfgNode(cfg)
        self.cfg = cfg
        self.num_masks = cfg.encoder_args.num_mask
        self.encoder = eval(cfg.encoder)(**cfg.encoder_args)
        self.aspp = ASPP(in_channel=cfg.aspp.in_channels, out_channel=cfg.aspp.out_channels)
        self.decoder = eval(cfg.decoder)(**cfg.decoder_args)
        self.loss_alpha_w = cfg.loss_alpha_w
        self.loss_alpha_lap_w = cfg.loss_alpha_lap_w
        self.loss_alpha_grad_w = cfg.loss_alpha_grad_w
        self.loss_atten_w = cfg.loss_atten_w
        self.reweight_os8 = cfg.loss_reweight_os8
        self.loss_dtSSD_w = cfg.loss_dtSSD_w
        for module in [self.aspp, self.decoder]:
            for name, p in module.named_parameters():
                if "context_token" in name:
                    continue
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)

    # ------------------------------------------------------------------------------------------------ forward
    def _sn_groups(self):
        """SpectralNorm modules split by stage: 'trunk' (encoder, ASPP, dense decoder stage) and 'detail' (the rest)."""
        groups = self.__dict__.get('_sn_groups_cache')
        if groups is None:
            from ..module.spectral_norm import SpectralNorm
            trunk_roots = [self.encoder, self.aspp] + list(self.decoder.dense_modules())
            trunk_ids = {id(m) for r in trunk_roots for m in r.modules() if isinstance(m, SpectralNorm)}
            allm = [m for m in self.modules() if isinstance(m, SpectralNorm)]
            # ordinary conv holders of the trunk ride along in the same batched kernels (layout / dtype conversion only)
            plain = list(self.aspp.plain_convs()) + list(self.decoder.plain_trunk_convs())
            enc_ids = {id(m) for m in self.encoder.modules() if isinstance(m, SpectralNorm)}
            groups = {'all': allm + plain, 'trunk': [m for m in allm if id(m) in trunk_ids] + plain,
                      'detail': [m for m in allm if id(m) not in trunk_ids],
                      # the trunk as two graphs (data parallel: the decoder's gradients are exchanged while the encoder's backward runs)
                      'enc': [m for m in allm if id(m) in enc_ids],
                      'dec': [m for m in allm if id(m) in trunk_ids and id(m) not in enc_ids] + plain}
            self.__dict__['_sn_groups_cache'] = groups
            self.__dict__['_sn_cache'] = {k: {} for k in groups}
        return groups

    def _prepare_spectral_norm(self, group='all'):
        """One batched HIP pipeline computes every spectrally-normalised weight of this forward (each SpectralNorm conv does
        exactly one power iteration per call; convs called more than once per forward fall back to the per-call kernels)."""
        mods = self._sn_groups()[group]
        MF.spectral_norm_prepare(mods, MF.compute_dtype(), self.__dict__['_sn_cache'][group])

    def _begin_step(self, device):
        """Per-forward bookkeeping done ONCE instead of per layer: zero the accumulator arena, bump every BatchNorm's
        num_batches_tracked with a single foreach op."""
        MF.ARENA.reset(device)
        # Everything cached by address (captured graphs, the BatchNorm counter list) is dropped when the parameters were
        # re-allocated: model.to(...) / .half() / .float() / a re-assigned .data
        # (every parameter and buffer, not a few sentinels: re-homing a subset -- FlatAdamW moves only the trainable ones, a per-layer
        # `p.data = ...` re-initialisation, SpectralNorm's fallback reassigning u / v -- must also drop the graphs; ~0.1 ms of host time)
        tensors = self.__dict__.get('_addr_tensors')
        if tensors is None:
            tensors = self.__dict__['_addr_tensors'] = list(self.parameters()) + list(self.buffers())
        sig = tuple([t.data_ptr() for t in tensors])
        if self.__dict__.get('_addr_sig') != sig:
            self.drop_graphs()
            self.__dict__.pop('_bn_counters', None)
            self.__dict__['_addr_sig'] = sig
        if self._defer_bn_counters():
            ctr = self.__dict__.get('_bn_counters')
            if ctr is None or any(c.device != device for c in ctr):
                ctr = [m.num_batches_tracked for m in self.modules()
                       if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None]
                self.__dict__['_bn_counters'] = ctr
            if ctr:
                torch._foreach_add_(ctr, 1)

    def _defer_bn_counters(self):
        # the temporal decoder calls some BatchNorm layers several times per forward (IMD smoothing convs, diff module):
        # their counters must advance per call, so only the image model batches the increments
        return self.training and not hasattr(self.decoder, 'os8_temp_module')

    def forward(self, batch, **kwargs):
        if not batch['image'].is_cuda:
            raise K.hip.MaggieHipError('MaGGIe (MI355X build) runs on the GPU only: move the batch to cuda (no CPU fallback)')
        self._begin_step(batch['image'].device)
        MF.DEFER_BN_COUNTERS = self._defer_bn_counters()
        MF.EAGER_TOKEN_CHECK = False                                  # read together with the other step flags in _forward_impl
        try:
            return self._forward_impl(batch, **kwargs)
        finally:
            MF.DEFER_BN_COUNTERS = False
            MF.EAGER_TOKEN_CHECK = True

    # ------------------------------------------------------------------------------------------------ dense trunk
    @_bn_counted
    def _trunk(self, geom, prepare_sn, x, enc_masks, masks, gt_alphas=None, mem_feat=None):
        """Static-shape part of the step: mask embedding + encoder + ASPP + dense decoder stage. Tensors in, tensors out."""
        b, n_f, n_i = geom
        if prepare_sn:
            self._prepare_spectral_norm('trunk')
        self.encoder.__dict__['defer_shortcuts'] = True           # (one graph holds encoder and decoder: the decoder may issue the fine shortcut branches)
        try:
            embedding, mid_fea = self.encoder(x, enc_masks)
        finally:
            self.encoder.__dict__['defer_shortcuts'] = False
        embedding = self.aspp(embedding)
        dense = self.decoder.dense_stage(embedding, mid_fea, b, n_f, n_i, masks, gt_alphas, mem_feat)
        MF.join_side()                                            # deferred shortcut branches ran on the side stream (functional.on_side_lane)
        return tuple(t for t in dense if t is not None)

    def _graph_policy(self):
        """hipGraph capture of the trunk: attribute `hip_graphs` (True/False) or env MAGGIE_HIP_GRAPHS (default on)."""
        flag = self.__dict__.get('hip_graphs')
        if flag is None:
            import os
            flag = os.environ.get('MAGGIE_HIP_GRAPHS', '1') != '0'
        if not flag:
            return False
        if self.training != torch.is_grad_enabled():
            return False
        if self.training and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and (torch.distributed.get_world_size() > 1 or MF.SYNCBN_WORLD1) and any(isinstance(m, nn.SyncBatchNorm) for m in self.modules()):
            # SyncBN exchanges batch statistics layer by layer (each layer's normalisation needs the global moments of ITS input, so the
            # ~142 small collectives of a step cannot be merged). MAGGIE_SYNCBN_GRAPHS=1 (opt-in: verified in a 1-rank process group only,
            # DESIGN.md section 6) records them into the hipGraphs -- through a PRIVATE RCCL communicator (maggie_amd/rccl_direct.py), not
            # through ProcessGroupNCCL, whose watchdog thread aborts the process when it queries an event recorded inside a capture.
            # Default since round 4 (the reference's target configs both set sync_bn: true, configs/maggie_image.yaml:33): the exchange is recorded
            # INTO the graphs -- mailbox kernels when every rank sits on one node with peer access, the private RCCL communicator otherwise
            # (parallel.syncbn_direct_comm). MAGGIE_SYNCBN_GRAPHS=0 asks for the eager path (host-launched torch.distributed collectives: 30 ms
            # against 12-13 ms per step in the 1-rank measurement); a SyncBatchNorm bound to a sub-group takes it, too (the communicator spans the
            # default group).
            import os
            sub = any(isinstance(m, nn.SyncBatchNorm) and m.process_group is not None and m.process_group is not torch.distributed.group.WORLD
                      for m in self.modules())
            if os.environ.get('MAGGIE_SYNCBN_GRAPHS', '1') == '0' or sub:
                if not self.__dict__.get('_syncbn_hint'):
                    self.__dict__['_syncbn_hint'] = True
                    logging.warning('MaGGIe (MI355X build): nn.SyncBatchNorm across ranks runs the step eagerly (142 host-launched collectives per step) -- %s',
                                    'a BatchNorm layer is bound to a sub-group' if sub else 'MAGGIE_SYNCBN_GRAPHS=0')
                return False
            from ... import parallel
            if parallel.syncbn_direct_comm() is None:             # collective on first use: every rank reaches its first training forward
                return False                                      # neither exchange form could be set up on this group (warned once): eager step
            return True
        return True

    @_bn_counted
    def _trunk_enc(self, prepare_sn, x, enc_masks):
        """First half of the trunk as its own graph: mask embedding + encoder -> (embedding, fea1..fea5)."""
        if prepare_sn:
            self._prepare_spectral_norm('enc')
        embedding, mid_fea = self.encoder(x, enc_masks)
        return (embedding,) + tuple(mid_fea['shortcut'])

    @_bn_counted
    def _trunk_dec(self, geom, prepare_sn, image_shape, embedding, f1, f2, f3, f4, f5, masks, gt_alphas=None):
        """Second half: ASPP + dense decoder stage (+ instance matte decoder) on the encoder graph's outputs."""
        b, n_f, n_i = geom
        if prepare_sn:
            self._prepare_spectral_norm('dec')
        embedding = self.aspp(embedding)
        mid_fea = {'shortcut': (f1, f2, f3, f4, f5), 'image': type('ImageShape', (), {'shape': image_shape})}
        dense = self.decoder.dense_stage(embedding, mid_fea, b, n_f, n_i, masks, gt_alphas, None)
        return tuple(t for t in dense if t is not None)

    def _split_trunk(self):
        """Trunk as two graph pairs (encoder | ASPP + dense decoder) instead of one: only useful for the overlapped gradient exchange of
        data-parallel runs (parallel.OverlappedGradSync). Attribute `split_trunk`, env MAGGIE_SPLIT_TRUNK=0/1, default: world size > 1."""
        flag = self.__dict__.get('split_trunk')
        if flag is None:
            import os
            env = os.environ.get('MAGGIE_SPLIT_TRUNK')
            if env is not None:
                flag = env != '0'
            else:
                flag = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        return bool(flag) and self.training

    def _rank_safe_graphs(self):
        """Must every rank replay the same graphs whatever its data? Attribute `rank_safe_graphs`, env MAGGIE_RANK_SAFE_GRAPHS=0/1, default:
        world size > 1 (the graphs then carry the overlapped gradient exchange, parallel.OverlappedGradSync)."""
        flag = self.__dict__.get('rank_safe_graphs')
        if flag is None:
            import os
            env = os.environ.get('MAGGIE_RANK_SAFE_GRAPHS')
            if env is not None:
                flag = env != '0'
            else:
                flag = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        return bool(flag)

    def drop_graphs(self):
        """Forget every captured hipGraph of this model (they are captured again at the second sight of their geometry). The device is drained
        first: destroying a graph executable -- its kernel-argument buffers and private memory pool -- while kernels of its last replay are still
        queued is a use-after-free on the device (round 6: one fatal memory fault in 13 runs of the test that swaps graph sets between steps)."""
        stores = [self.__dict__.get(st) for st in ('_trunk_graphs', '_trunk_enc_graphs', '_detail_graphs', '_detail_names')]
        if any(stores) and torch.cuda.is_available():
            torch.cuda.synchronize()
        for st in stores:
            if st:
                st.clear()

    def _graphed(self, store, key, fn, inputs, grad_inputs=()):
        """fn(*inputs) eagerly the first time `key` is seen, captured into hipGraphs (forward + backward) the second time, replayed from
        then on. -> (outputs, replayed?). `key` None: always eager."""
        graphs = self.__dict__.setdefault(store, {})
        entry = graphs.get(key) if key is not None else None
        if isinstance(entry, int) and entry >= 1:
            entry = self._capture(fn, inputs, grad_inputs)
            graphs.pop(key, None)
            graphs[key] = entry                                   # most recently used = last
            # bounded: each captured graph pins its own activation pool. Evict the least recently USED graph (never the one just
            # captured); sighting counters of geometries that were never captured are bounded separately
            captured = [k for k, v in graphs.items() if not isinstance(v, int)]
            if len(captured) > 4:
                torch.cuda.synchronize()                          # (see drop_graphs: never destroy a graph the device may still be executing)
            for k in captured[:max(0, len(captured) - 4)]:
                graphs.pop(k)
            counters = [k for k, v in graphs.items() if isinstance(v, int)]
            for k in counters[:max(0, len(counters) - 16)]:
                graphs.pop(k)
        if entry is None or isinstance(entry, int) or entry == 'failed':
            if key is not None and entry != 'failed':
                graphs[key] = (entry or 0) + 1
            return None, False
        graphs[key] = graphs.pop(key)                             # LRU order: a replayed graph moves to the end
        return entry(*inputs), True

    def _capture(self, fn, inputs, grad_inputs=()):
        from ... import graphs
        mutable = [t for t in self.buffers()] + [p for p in self.parameters() if not p.requires_grad] + self.decoder.head_state()
        try:
            amp_dt = (torch.get_autocast_dtype('cuda') if hasattr(torch, 'get_autocast_dtype') else torch.get_autocast_gpu_dtype()) \
                if torch.is_autocast_enabled() else torch.bfloat16
            with torch.autocast('cuda', dtype=amp_dt, enabled=torch.is_autocast_enabled(), cache_enabled=False):
                g = graphs.GraphedCallable(fn, inputs, self, mutable, self.training, grad_inputs=grad_inputs,
                                           grad_sink=self.__dict__.get('grad_sink'))
            overlap = self.__dict__.get('_grad_overlap')
            if overlap is not None:
                g.grad_hook = overlap.reduce_async
            return g
        except Exception as e:
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                raise                  # data parallel: every rank must take the same path (same collectives in the same order)
            logging.warning('hipGraph capture of a MaGGIe stage failed (%s: %s); staying eager', type(e).__name__, e)
            return 'failed'

    def _run_trunk(self, geom, x, enc_masks, masks, gt_alphas, mem_feat):
        """Eager the first time a batch geometry is seen, captured into hipGraphs from the second time on."""
        extra = [gt_alphas] if (self.training and gt_alphas is not None) else []
        inputs = [x, enc_masks, masks] + extra
        has_hidden = hasattr(self.decoder, 'os8_temp_module')
        key, from_graph = None, False
        if mem_feat is None and self._graph_policy():
            key = (geom, self.training, MF.compute_dtype(), tuple((tuple(t.shape), t.dtype) for t in inputs))
        if key is not None and self._split_trunk():
            enc, r1 = self._graphed('_trunk_enc_graphs', key, lambda *t: self._trunk_enc(True, *t), [x, enc_masks])
            if r1:
                dec_in = list(enc) + [masks] + extra
                out, r2 = self._graphed('_trunk_graphs', ('dec',) + key, lambda *t: self._trunk_dec(geom, True, x.shape, *t), dec_in,
                                        grad_inputs=range(len(enc)))
                from_graph = r2
                if not r2:                                        # decoder graph not there yet: run it eagerly on the encoder graph's outputs
                    self._prepare_spectral_norm('dec')
                    out = self._trunk_dec(geom, False, x.shape, *dec_in)
                self._prepare_spectral_norm('detail')
            else:
                self.__dict__.setdefault('_trunk_graphs', {}).setdefault(('dec',) + key, 0)
                g = self.__dict__['_trunk_graphs']
                if isinstance(g[('dec',) + key], int):
                    g[('dec',) + key] += 1                        # both halves are captured at the same (second) sight of the geometry
                self._prepare_spectral_norm('all')
                out = self._trunk(geom, False, *inputs, mem_feat=mem_feat)
        else:
            out, replayed = self._graphed('_trunk_graphs', key, lambda *t: self._trunk(geom, True, *t), inputs)
            from_graph = replayed
            if not replayed:
                self._prepare_spectral_norm('all')
                out = self._trunk(geom, False, *inputs, mem_feat=mem_feat)
            else:
                self._prepare_spectral_norm('detail')
        if not isinstance(out, list):
            out = list(out)
        self.__dict__['_trunk_replayed'] = bool(from_graph)       # _run_detail: outputs of an EAGER detail stage may alias the trunk graph's buffers
        if not has_hidden:
            out.insert(4, None)
        return tuple(out)

    def _forward_impl(self, batch, **kwargs):
        masks, alphas, trans_gt, b, n_f, h, w, n_i, chosen_ids, x, enc_masks = self.forward_inputs(batch)
        dense, nonzero = self.decoder.split_dense_flag(self._run_trunk((b, n_f, n_i), x, enc_masks, masks, alphas, kwargs.get('mem_feat')))
        # ---- the ONE device->host read of the step: NaN instance tokens (mask_attention.py:95-98 raises) and, in training, whether the
        # coarse alpha is identically zero (resnet_inst_matt_spconv.py:314: the ground truth then guides the detail region). Everything
        # else the detail stage needs from the host is drawn below, in the reference's order; no site count is ever read back.
        x_os8 = dense[0]
        auto_cap = self.decoder.sparse_capacity() == 'auto'
        ovf = self.decoder.__dict__.get('_sparse_overflow') if self.decoder.sparse_bounded() else None
        comm = None
        if self.training:
            from ... import parallel as _par
            comm = _par.SYNCBN_COMM if hasattr(_par.SYNCBN_COMM, 'error_word') else None
        tok = dense[2]
        if tok.is_cuda and tok.dtype == torch.float32 and tok.is_contiguous():
            # [NaN tokens, coarse alpha all zero (written by the up-sampling kernel), sparse overflow (sticky: raised by an EARLIER step's detail stage,
            # sparse_head.DeviceLevel), mailbox error word (a SyncBatchNorm peer that did not arrive leaves 1 + its rank)] in one launch
            # (mg_step_flags; was isnan + any + compare(s) + stack in front of the copy the host waits for)
            # ('auto' sparse capacity: the flag words are the head of a persistent int32 [8] whose tail the previous detail stage filled with its live
            # site counts -- the same single copy brings them along)
            words = self.decoder.sparse_flag_words(tok.device) if auto_cap else torch.empty(4, dtype=torch.int32, device=tok.device)
            K.hip.call('mg_step_flags', K.hip.ptr(tok), K.c_long(tok.numel()), K.hip.ptr(nonzero if self.training else None), K.hip.ptr(ovf),
                       K.hip.ptr(None if comm is None else comm.error_word), K.hip.ptr(words), K.hip.stream())
            words = words.tolist()
            if auto_cap:
                self.decoder.sparse_note_counts(words[4:8], bool(ovf is not None and words[2]))
                if ovf is not None and words[2]:
                    ovf.zero_()
                    words[2] = 0
            flags = [bool(words[0])] + ([bool(words[1])] if self.training else []) + ([bool(words[2])] if ovf is not None else [])
            word = words[3]
        else:
            fl = [torch.isnan(tok).any()] + ([nonzero[0] == 0] if self.training else [])
            if auto_cap:
                # (fp16 / bf16 tokens: the generic path) counts and overflow of the previous detail stage in their own small read
                prev = self.decoder.sparse_flag_words(tok.device)[4:8].tolist()
                over = bool(ovf is not None and int(ovf[0]) != 0)
                self.decoder.sparse_note_counts(prev, over)
                if over:
                    ovf.zero_()
                ovf = None
            if ovf is not None:
                fl.append(ovf[0] != 0)
            if comm is not None:
                fl = [f.to(torch.int32) for f in fl] + [comm.error_word[0]]
            flags = torch.stack(fl).tolist()
            word = flags.pop() if comm is not None else 0
        if comm is not None and word:
            comm.error_word.zero_()
            comm.raise_for(word)
        if flags[0]:
            raise ValueError("Mask is empty")
        if ovf is not None and flags[-1]:
            ovf.zero_()
            raise K.hip.MaggieHipError(
                'MaGGIe (MI355X build): the detail region of a previous step had more active sites than the sparse head was sized for '
                '(sparse_capacity = %.2f of all sites); the sites beyond the capacity were dropped, i.e. that step\'s refinement and gradients are '
                'incomplete. Raise model.decoder.sparse_capacity_frac / MAGGIE_SPARSE_CAPACITY (1.0 cannot overflow; \'auto\' follows the workload).' % self.decoder.sparse_capacity())
        P = b * n_f * (x_os8.shape[1] if self.training else n_i)
        self.decoder.__dict__['_sparse_last_key'] = (P, h, w, bool(self.training))     # whose site counts the NEXT step's flag read brings ('auto' capacity)
        plan = self.decoder.detail_plan(batch.get('iter', 0), bool(self.training and flags[1]), P, x.device)
        use_fuse_w = bool(self.training and np.random.rand() < 0.75)            # arch/maggie.py:99-101
        names, tensors = self._run_detail(dense, (b, n_f, n_i, h, w), plan, use_fuse_w, alphas, trans_gt)
        res = dict(zip(names, tensors))
        output = {k[4:]: v for k, v in res.items() if k.startswith('out/')}
        if self.training:
            loss_dict = {k[5:]: v for k, v in res.items() if k.startswith('loss/')}
            if chosen_ids is not None:
                for k, v in output.items():
                    output[k] = v[:, :, chosen_ids, :, :]
            return output, loss_dict
        return output

    # ------------------------------------------------------------------------------------------------ detail stage + losses
    @_bn_counted
    def _detail_and_loss(self, geom, plan, n_dense, *tensors):
        """Static-shape tail of the step as a function of tensors only: detail region -> sparse refinement -> fusion -> output dict
        (-> losses). `tensors` = the trunk's outputs, then [ground-truth alphas, transition maps, fuse-weight flag] in training.
        -> (names, tensors): 'out/<key>' entries (detached) and, in training, 'loss/<name>' entries ('loss/total' differentiable)."""
        b, n_f, n_i, h, w = geom
        dense = list(tensors[:n_dense])
        if not hasattr(self.decoder, 'os8_temp_module'):
            dense.insert(4, None)                                 # the image decoder has no recurrent hidden state
        alphas = trans_gt = use_w = None
        if self.training:
            alphas, trans_gt, step_flags = tensors[n_dense:n_dense + 3]      # step_flags: int32 [use the fuse weights, ground truth guides the region]
            use_w = step_flags[:1]
            if plan.get('use_gt') is None:
                plan = dict(plan, use_gt_dev=step_flags[1:2])
        pred = self.decoder.detail_stage(dense, (h, w), b, n_f, n_i, plan, alphas, spar_gt=trans_gt)
        alpha_pred = pred.pop("refined_masks")
        detail_bits = pred.pop('detail_bits', None)
        if self.training and 'weight_os4_bits' in pred and detail_bits is not None:
            # `np.random.rand() < 0.75` (arch/maggie.py:99-101) was drawn by the caller: a device flag selects the weight planes, so one
            # captured graph serves both outcomes -- on the BIT planes, unpacked once straight to fp32
            pick = use_w.bool()
            shape = pred['detail_mask'].shape
            weight_os4 = K.bits_unpack_f32(torch.where(pick, pred.pop('weight_os4_bits'), detail_bits), w, shape)
            weight_os1 = K.bits_unpack_f32(torch.where(pick, pred.pop('weight_os1_bits'), detail_bits), w, shape)
        else:
            pred.pop('weight_os4_bits', None), pred.pop('weight_os1_bits', None)
            weight_os4 = pred["detail_mask"].type(alpha_pred.dtype)
            weight_os1 = weight_os4
        output = self.transform_output(b, n_f, h, w, n_i, pred, alpha_pred)
        names, outs = [], []
        if self.training:
            alphas = alphas.view(-1, n_i, h, w)
            trans_gt = trans_gt.view(-1, n_i, h, w)
            # `pred[k] = v * valid_masks` (:112-118; valid = the plane has a ground-truth transition region). Where only the fused matting
            # losses read the three alpha planes (image model), the per-plane 0 / 1 factor is handed to the loss kernels instead: no
            # reduction over trans_gt, no three multiplies over (N, 10, H, W) planes forward and backward.
            fold = self.loss_dtSSD_w <= 0 and n_i == self.num_masks and trans_gt.dtype == torch.float32
            pvalid = K.plane_flags(trans_gt.contiguous()) if fold else None
            valid_masks = None if fold else (trans_gt.sum((2, 3), keepdim=True) > 0).float()
            for k, v in list(pred.items()):
                if 'loss' in k or 'mem_' in k:
                    continue
                if k in ('detail_mask', 'weight_os4', 'weight_os1'):
                    continue        # the reference multiplies these too (:114-117) but never reads them again
                if fold and k in ('alpha_os1', 'alpha_os4', 'alpha_os8'):
                    continue
                if valid_masks is None:
                    valid_masks = pvalid.view(-1, n_i, 1, 1).float()
                pred[k] = v * valid_masks
            loss_dict = self.compute_loss(pred, weight_os4, weight_os1, alphas, trans_gt, (b, n_f, self.num_masks, h, w),
                                          reweight_os8=self.reweight_os8, defer_total=True, pvalid=pvalid)
            self.update_additional_decoder_loss(pred, loss_dict)
            self._finish_total(loss_dict)
            for k, v in loss_dict.items():
                names.append('loss/' + k)
                outs.append(v if k == 'total' else v.detach())          # the caller back-propagates loss['total'] (engine/train.py:265-268)
        else:
            for k, v in output.items():
                output[k] = v[:, :, :n_i]
            for k in pred:
                if k.startswith("mem_"):
                    output[k] = pred[k]
        for k, v in output.items():
            names.append('out/' + k)
            outs.append(v.detach())
        return names, tuple(outs)

    def _run_detail(self, dense, geom, plan, use_fuse_w, alphas, trans_gt):
        """The detail stage eagerly the first time a (geometry, mode) is seen, from its own pair of hipGraphs afterwards; its differentiable
        inputs are the trunk's outputs, so backward chains detail graph -> trunk graph."""
        dense = [t for t in dense if t is not None]
        n_dense = len(dense)
        extra = []
        rank_safe = self.training and self._rank_safe_graphs()
        if self.training:
            extra = [alphas, trans_gt, torch.tensor([int(use_fuse_w), int(plan['use_gt'])], dtype=torch.int32).to(dense[0].device, non_blocking=True)]
        host_w = [plan['widths']] if plan['widths'] is not None else []
        inputs = list(dense) + extra + host_w
        # rank_safe: `use_gt` is per-rank data (random.random(), x_os8.sum() == 0): as part of the static plan / graph key it would let the
        # ranks of a data-parallel job replay DIFFERENT graphs -- i.e. issue different gradient collectives. It travels as a device flag then.
        static_plan = {'use_gt': None if rank_safe else plan['use_gt'], 'with_atten': plan['with_atten']}

        def fn(*t):
            p = dict(static_plan, widths=t[n_dense + len(extra)] if host_w else None)
            names, outs = self._detail_and_loss(geom, p, n_dense, *t)
            fn.names = names
            return outs

        key = self._detail_key(geom, static_plan, inputs) if self._graph_policy() else None
        names_key = ('names', key)
        outs, replayed = self._graphed('_detail_graphs', key, fn, inputs, grad_inputs=range(n_dense))
        store = self.__dict__.setdefault('_detail_names', {})
        if replayed:
            return store[names_key], self._own_outputs(store[names_key], outs)
        outs = fn(*inputs)
        store[names_key] = fn.names
        if self.__dict__.get('_trunk_replayed'):
            # eager detail stage on a replayed trunk (the first two sightings of a geometry): what it passes through (alpha_os8, the recurrent
            # feature) is the trunk graph's own memory -- the caller gets private copies, as from a replayed detail graph
            outs = tuple(o.clone() if (getattr(o, '_mg_static', False) or any(o.data_ptr() == d.data_ptr() for d in dense)) else o for o in outs)
        return fn.names, outs

    def _detail_key(self, geom, static_plan, inputs):
        """What selects a captured detail graph: geometry, mode, dtype, the STATIC part of the plan and the input signature -- never per-rank
        data when `static_plan['use_gt']` is None (rank-safe mode: the guidance choice is a device flag among `inputs`)."""
        return (geom, self.training, MF.compute_dtype(), static_plan['use_gt'], static_plan['with_atten'],
                tuple((tuple(t.shape), t.dtype) for t in inputs), self.decoder.sparse_caps_version())      # (row capacities are baked into a capture)

    @staticmethod
    def _own_outputs(names, outs):
        """A replayed graph hands back views of ITS static output buffers, which the next forward with the same geometry overwrites.
        Everything the caller receives (alphas, detail mask, memory features, logged loss scalars) is copied out with one multi-tensor
        launch, so outputs held across calls (VideoWindow, metric / demo code) keep their values -- as they do on the eager path."""
        outs = list(outs)
        idx = [i for i, (n, t) in enumerate(zip(names, outs)) if not t.requires_grad]
        # the logged loss scalars (a dozen one-element fp32 tensors): ONE concatenation gives them fresh memory, views are handed out
        small = [i for i in idx if outs[i].numel() == 1 and outs[i].dtype == torch.float32]
        if len(small) > 1:
            pack = torch.cat([outs[i].reshape(1) for i in small])
            for j, i in enumerate(small):
                outs[i] = pack[j].view(outs[i].shape)
            idx = [i for i in idx if i not in set(small)]
        if idx:
            fresh = [torch.empty_like(outs[i]) for i in idx]
            MF.K.copy_k(fresh, [outs[i] for i in idx])            # one launch for the four 42 MB alpha planes + the index map (was one copy kernel each)
            for i, f in zip(idx, fresh):
                outs[i] = f
        return tuple(t.clone() if t.requires_grad else t for t in outs)        # 'loss/total': a differentiable one-element copy

    @staticmethod
    def _add_to_total(loss_dict, term, coef):
        if '_total_terms' in loss_dict:
            loss_dict['_total_terms'][0].append(term)
            loss_dict['_total_terms'][1].append(coef)
        else:
            loss_dict['total'] = loss_dict['total'] + term * coef

    @staticmethod
    def _finish_total(loss_dict):
        terms, coefs = loss_dict.pop('_total_terms')
        loss_dict['total'] = MF.scalar_lincomb(terms, coefs) if terms else 0

    def update_additional_decoder_loss(self, pred, loss_dict):
        if 'loss_max_atten' in pred and self.loss_atten_w > 0:
            loss_dict['loss_max_atten'] = pred['loss_max_atten']
            self._add_to_total(loss_dict, loss_dict['loss_max_atten'], self.loss_atten_w)

    def transform_output(self, b, n_f, h, w, n_i, pred, alpha_pred):
        output = {}
        n_out = self.num_masks if (self.training and self.num_masks > 0) else n_i
        if 'alpha_os1' in pred:
            output['alpha_os1'] = pred['alpha_os1'][:, :n_out].reshape(b, n_f, n_out, h, w)
            output['alpha_os4'] = pred['alpha_os4'][:, :n_out].reshape(b, n_f, n_out, h, w)
        output['alpha_os8'] = pred['alpha_os8'][:, :n_out].reshape(b, n_f, n_out, h, w)
        output['refined_masks'] = alpha_pred[:, :n_out].reshape(b, n_f, n_out, h, w)
        if 'detail_mask' in pred:
            output['detail_mask'] = pred['detail_mask'][:, :n_out].reshape(b, n_f, n_out, h, w)
        return output

    def forward_inputs(self, batch):
        """Input plumbing of arch/maggie.py:160-198 up to (not including) the encoder call."""
        x = batch['image']
        masks = batch['mask']
        alphas = batch.get('alpha', None)
        trans_gt = batch.get('transition', None)
        b, n_f, _, h, w = x.shape
        n_i = masks.shape[2]
        if not x.is_cuda:
            raise K.hip.MaggieHipError('MaGGIe (MI355X build) runs on the GPU only: move the batch to cuda (no CPU fallback)')
        x = x.reshape(-1, 3, h, w).float()
        masks_lr = masks.flatten(0, 1).float()                      # kept at its own resolution for the packing kernel
        hm, wm = masks.shape[-2:]
        if wm == w and hm == h:
            masks = masks_lr
        elif w % wm == 0 and h % hm == 0 and w // wm == h // hm and 8 % (w // wm) == 0:
            # The reference up-scales the guidance masks to (h, w) (nearest, arch/maggie.py:176-178) only to reduce them again: the instance
            # matte decoder average-pools them to OS8 and thresholds (> 0), `valid_masks` is (sum > 0) per plane. For an integer down-scale
            # that divides 8 both are functions of the low-resolution mask alone (every OS8 cell is a whole number of mask pixels), so the
            # 40 MB full-resolution copy, its reduction and its pooling are skipped: the decoder receives the mask as it came.
            masks = masks_lr
        else:
            masks = F.interpolate(masks_lr, size=(h, w), mode="nearest")
        masks, alphas, trans_gt, n_i, chosen_ids, enc_masks = self.prepare_input(x, masks, masks_lr, alphas, trans_gt, b, n_f, h, w, n_i)
        if alphas is not None:
            alphas = alphas.reshape(-1, n_i, h, w)
        if trans_gt is not None:
            trans_gt = trans_gt.reshape(-1, n_i, h, w)
        return masks, alphas, trans_gt, b, n_f, h, w, n_i, chosen_ids, x.contiguous(), enc_masks.contiguous()

    def forward_encoder(self, batch):
        """Same return tuple as the reference's forward_encoder (arch/maggie.py:160-198)."""
        masks, alphas, trans_gt, b, n_f, h, w, n_i, chosen_ids, x, enc_masks = self.forward_inputs(batch)
        self._prepare_spectral_norm('all')
        embedding, mid_fea = self.encoder(x, enc_masks)
        embedding = self.aspp(embedding)
        return masks, alphas, trans_gt, b, n_f, h, w, n_i, chosen_ids, embedding, mid_fea

    def prepare_input(self, x, masks, masks_lr, alphas, trans_gt, b, n_f, h, w, n_i):
        """arch/maggie.py:200-235: pad the guidance masks to `num_mask` channels (eval: zeros at the end; train: random
        slots via np.random.choice). Returns the (possibly re-slotted) masks and the low-resolution masks for the encoder."""
        chosen_ids = None
        enc_masks = masks_lr
        if self.num_masks - n_i > 0:
            hm, wm = masks_lr.shape[-2:]
            if not self.training:
                enc_masks = torch.cat([masks_lr, masks_lr.new_zeros((b * n_f, self.num_masks - n_i, hm, wm))], dim=1)
            else:
                chosen_ids = np.random.choice(self.num_masks, n_i, replace=False)
                enc_masks = masks_lr.new_zeros((b * n_f, self.num_masks, hm, wm))
                enc_masks[:, chosen_ids] = masks_lr
                new_masks = masks.new_zeros((b * n_f, self.num_masks, *masks.shape[-2:]))
                new_masks[:, chosen_ids] = masks
                masks = new_masks
                if alphas is not None:
                    na = alphas.new_zeros((b, n_f, self.num_masks, h, w))
                    na[:, :, chosen_ids] = alphas
                    alphas = na
                if trans_gt is not None:
                    nt = trans_gt.new_zeros((b, n_f, self.num_masks, h, w))
                    nt[:, :, chosen_ids] = trans_gt
                    trans_gt = nt
                n_i = self.num_masks
        return masks, alphas, trans_gt, n_i, chosen_ids, enc_masks

    # ------------------------------------------------------------------------------------------------ losses
    def compute_loss(self, pred, weight_os4, weight_os1, alphas, trans_gt, alpha_shape, reweight_os8=True, defer_total=False, pvalid=None):
        """arch/maggie.py:268-368: weighted L1 + Laplacian-pyramid L1 + Sobel-gradient L1 at OS1 (x2) / OS4 / OS8 (+ dtSSD for video), same
        loss names. Every term is a fused HIP pipeline (csrc/losses.hip) -- there is no torch fallback: what the kernels do not cover is
        rejected loudly (`loss_alpha_type` other than the 'l1' of maggie_{image,video}.yaml; sizes that are not multiples of 8)."""
        a1, a4, a8 = pred.get('alpha_os1', None), pred.get('alpha_os4', None), pred['alpha_os8']
        lt = self.cfg.loss_alpha_type
        if lt not in ('l1', 'l2'):
            raise NotImplementedError("NotImplemented loss type {}".format(lt))          # arch/maggie.py:253,266
        if lt != 'l1' or a1 is None or a8.shape[-1] % 8 or a8.shape[-2] % 8 or not alphas.is_cuda:
            raise K.hip.MaggieHipError(
                "MaGGIe (MI355X build): the fused HIP loss kernels implement loss_alpha_type='l1' (configs/maggie_{image,video}.yaml) on "
                "CUDA planes whose height and width are multiples of 8; got type %r, size %dx%d. There is no torch fallback."
                % (lt, a8.shape[-2], a8.shape[-1]))
        loss_dict = {}
        alphas = alphas.float()
        weight_os8 = MF.os8_weight(alphas, a8.float(), reweight_os8, pvalid)     # arch/maggie.py:271-281 in one pass (mg_os8_weight)
        n_i = alphas.shape[1]
        if self.num_masks - n_i > 0:
            padding = torch.zeros((alphas.shape[0], self.num_masks - n_i, *alphas.shape[-2:]), device=alphas.device)
            alphas = torch.cat([alphas, padding], dim=1)
            trans_gt = torch.cat([trans_gt, padding], dim=1)
        # one fused HIP pipeline per scale: weighted L1 + Laplacian-pyramid L1 + Sobel-gradient L1 (fwd sums + exact bwd)
        # the three scales share shape and target: ONE batched pipeline (9 launches forward, 9 backward, whatever the number of scales)
        (r1, l1_, g1), (r4, l4_, g4), (r8, l8_, g8) = MF.matting_losses_multi([a1, a4, a8], alphas, [weight_os1, weight_os4, weight_os8], pvalid)
        # The sums of arch/maggie.py:283-300 (loss_x = 2 * os1 + os4 + os8, total = sum_x w_x * loss_x): the per-family values are for the log
        # (no gradient flows through them -- only through `total`), `total` is ONE weighted sum of the nine (twelve) scale terms
        # (mg_scalar_lincomb: one launch forward, one backward, instead of ~12 + ~14 one-element torch kernels)
        terms, coefs = [], []

        def family(name, w, t1, t4, t8):
            loss_dict[name + '_os1'], loss_dict[name + '_os4'], loss_dict[name + '_os8'] = t1, t4, t8
            with torch.no_grad():
                loss_dict[name] = MF.scalar_lincomb([t1, t4, t8], [2.0, 1.0, 1.0])
            terms.extend([t1, t4, t8])
            coefs.extend([2.0 * w, w, w])

        if self.loss_alpha_w > 0:
            family('loss_rec', self.loss_alpha_w, r1, r4, r8)
        if self.loss_alpha_lap_w > 0:
            family('loss_lap', self.loss_alpha_lap_w, l1_, l4_, l8_)
        if self.loss_alpha_grad_w > 0:
            family('loss_grad', self.loss_alpha_grad_w, g1, g4, g8)
        if self.loss_dtSSD_w > 0:
            rs = lambda t: t.reshape(*alpha_shape)
            d1 = MF.dtssd_loss(rs(a1), rs(alphas), rs(weight_os1))       # loss.py:7-16 as a fused HIP reduction (mg_dtssd_fwd / _bwd)
            d4 = MF.dtssd_loss(rs(a4), rs(alphas), rs(weight_os4))
            d8 = MF.dtssd_loss(rs(a8), rs(alphas), rs(weight_os8))
            family('loss_dtSSD', self.loss_dtSSD_w, d1, d4, d8)
        if len(terms) == 9 and self.loss_dtSSD_w <= 0:
            # scale-major order [rec, lap, grad of OS1 | of OS4 | of OS8] -- the order the fused loss pipeline holds its nine sums in: the gradients of
            # the weighted sum then ARE that tensor's gradient, in place (functional.SplitGrid; family-major order cost four stack launches per step)
            order = [3 * j + s_ for s_ in range(3) for j in range(3)]
            terms, coefs = [terms[i] for i in order], [coefs[i] for i in order]
        if defer_total:
            loss_dict['_total_terms'] = (terms, coefs)       # update_additional_decoder_loss appends the decoder's terms; _finish_total sums once
        else:
            loss_dict['total'] = MF.scalar_lincomb(terms, coefs) if terms else 0
        return loss_dict
