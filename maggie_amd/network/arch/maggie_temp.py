"""MaGGIe_Temp -- the video model (maggie/network/arch/maggie_temp.py:5-79): MaGGIe plus the temporal decoder's extra outputs and
losses, and at inference the alpha-level aggregation over a 3-frame window (t-1, t, t+1), here ONE in-place HIP kernel
(mg_temporal_fuse, maggie_amd/csrc/temporal.hip) instead of ~15 elementwise launches."""
from ... import kernels as K
from .maggie import MaGGIe

# decoder entry -> output name of the temporal extras (transform_output) and decoder loss -> loss_dict name
_EXTRA_OUTPUTS = (('diff_backward', 'diff_pred_backward'), ('diff_forward', 'diff_pred_forward'))
_EXTRA_LOSSES = ('loss_temp_fusion', 'loss_temp_dtssd')


class MaGGIe_Temp(MaGGIe):
    def transform_output(self, b, n_f, h, w, n_i, pred, alpha_pred):
        output = super().transform_output(b, n_f, h, w, n_i, pred, alpha_pred)
        extras = {name: pred.pop(key, None) for key, name in _EXTRA_OUTPUTS}
        temp_alpha = pred.pop('temp_alpha', None)
        if extras['diff_pred_backward'] is not None:
            for name, diff in extras.items():                                   # one difference map per frame, shared by the instances
                output[name] = diff.repeat(1, 1, n_i, 1, 1)
            output['temp_alpha'] = temp_alpha
        return output

    def update_additional_decoder_loss(self, pred, loss_dict):
        super().update_additional_decoder_loss(pred, loss_dict)
        if 'loss_temp' in pred:                                                 # BCE + dtSSD on the difference maps, already weighted
            loss_dict['loss_temp_bce'], loss_dict['loss_temp'] = pred['loss_temp_bce'], pred['loss_temp']
            self._add_to_total(loss_dict, pred['loss_temp'], 1.0)
        loss_dict.update({k: pred[k] for k in _EXTRA_LOSSES if k in pred})

    def forward(self, batch, **kwargs):
        output = super().forward(batch, **kwargs)
        if self.training:
            return output
        # window aggregation (maggie_temp.py:34-77): frames 1 and 2 of refined_masks are rewritten in place; "t+1" is the clip's LAST frame
        # (alphas[:, -1], :48) -- frame 2 for the 3-frame evaluation window of engine/test.py, any n_f >= 3 works like the reference
        alphas = output['refined_masks']                                        # (1, n_f, n_i, H, W)
        assert alphas.shape[0] == 1 and alphas.shape[1] >= 3, 'the eval-time aggregation works on one window of >= 3 frames'
        prev = kwargs.get('prev_pred')                                          # fused t-1 of the previous window, else frame 0
        if prev is not None:
            prev = prev.to(alphas.device).float().contiguous()
        fused = alphas[0].float().contiguous()
        K.temporal_fuse_(fused, prev, output['diff_pred_forward'][0].float().contiguous(),
                         output['diff_pred_backward'][0].float().contiguous())
        alphas[0, 1:3] = fused[1:3]
        return output
