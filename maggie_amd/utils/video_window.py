"""Streaming video evaluation window on the device (second half) -- the bookkeeping of `eval_video`
(maggie/engine/test.py:237-286) without the `.cpu().numpy()` hop after every forward.

The video model is run on overlapping 3-frame clips (t-1, t, t+1). The reference keeps numpy arrays of predictions, ground truths and
trimaps; per clip it (1) replaces the previous clip's t+1 prediction by the new clip's t and appends the new t+1 (:237-249), (2) hands the
oldest frame (all remaining frames on the last clip) to the saving callback (:256-260), (3) evaluates the non-temporal metrics on frame
t-1 (t-1..t+1 on the last clip) and the temporal ones on the two frames before, when there are any (:266-274), (4) keeps the last three
frames (:282-286). `VideoWindow.push` does the same with device tensors and returns views for the callback / the device metrics
(maggie_amd.utils.metric)."""
import torch


class VideoWindow:
    def __init__(self):
        self.reset()

    def reset(self):
        self.preds = self.gts = self.trimaps = None
        self.names = []

    def push(self, alpha, alpha_gt, trimap, names, is_first, is_last):
        """alpha, alpha_gt, trimap: (1, 3, n_i, H, W) tensors of one clip (any device); names: its 3 frame names.
        -> dict: 'save' = (names, preds) for the saving callback; 'current' = (preds, trimaps, gts) for the non-temporal metrics;
        'previous' = the same triple for the temporal metrics or None."""
        if is_first:
            self.reset()
            self.preds, self.gts, self.trimaps, self.names = alpha[0], alpha_gt[0], trimap[0], list(names)
        else:
            self.gts = torch.cat([self.gts, alpha_gt[0, 2:]], 0)                 # t+1 joins the cumulative GTs / trimaps (:243-245)
            self.trimaps = torch.cat([self.trimaps, trimap[0, 2:]], 0)
            self.names = self.names + list(names[2:])
            self.preds = torch.cat([self.preds[:-1], alpha[0, 1:]], 0)           # drop the old t+1, add the new t and t+1 (:248)
        n = self.preds.shape[0]
        end_idx = 1 if not is_last else n
        out = {'save': (self.names[:end_idx], self.preds[None, :end_idx])}
        previous = None
        if n > 3:
            # (:266-270) the reference slices [-4:-3] while the clip is not the last one; on the last clip its end index is
            # `len(prev_preds)` of the PREVIOUS iteration (= 1 whenever that iteration had 4 stored frames), i.e. [-4:1] of 4 frames:
            # the same single frame. (On a video of exactly two clips the reference raises -- len(None) -- and a single
            # first-and-last clip has n == 3; both corner cases get the single frame / None here.)
            previous = (self.preds[-4:-3], self.trimaps[-4:-3], self.gts[-4:-3])
        end_all = -2 if not is_last else n
        out['current'] = (self.preds[-3:end_all], self.trimaps[-3:end_all], self.gts[-3:end_all])
        out['previous'] = previous
        if n > 3:                                                                # keep the last three (:282-286)
            self.preds, self.gts, self.trimaps, self.names = self.preds[-3:], self.gts[-3:], self.trimaps[-3:], self.names[-3:]
        return out
