"""TEST INFRASTRUCTURE ONLY. numpy restatement of the window bookkeeping of eval_video, maggie/engine/test.py:237-286, statement by
statement (lists/arrays exactly as there). `end_pred_idx` on the last clip is `len(prev_preds)` of the PREVIOUS loop iteration, as executed by
the reference: 1 when that iteration stored 4 frames (so [-4:1] of 4 frames = the same single frame as [-4:-3]); the reference raises on a
two-clip video (len(None)) -- this restatement keeps the single frame there. Parity unpinned: eval_video needs a model and a data loader to run."""
import numpy as np


class Window:
    def __init__(self):
        self.all_preds, self.all_gts, self.all_trimap, self.all_image_names = [], [], [], []

    def push(self, alpha, alpha_gt, trimap, image_names, is_first, is_last):
        if is_first:
            self._prev_len = None
            self.all_preds, self.all_gts, self.all_trimap, self.all_image_names = alpha[0], alpha_gt[0], trimap[0], list(image_names)
        else:
            self.all_gts = np.concatenate([self.all_gts, alpha_gt[0, 2:]], axis=0)
            self.all_trimap = np.concatenate([self.all_trimap, trimap[0, 2:]], axis=0)
            self.all_image_names = self.all_image_names + list(image_names[2:])
            self.all_preds = np.concatenate([self.all_preds[:-1], alpha[0, 1:]], axis=0)
        all_preds = self.all_preds
        end_idx = 1 if not is_last else len(all_preds)
        save = (self.all_image_names[:end_idx], all_preds[None, :end_idx])
        prev_len = getattr(self, '_prev_len', None)                       # len(prev_preds) left over from the previous iteration
        end_pred_idx = -3 if (not is_last or prev_len is None) else prev_len
        prev = None
        if len(all_preds) > 3:
            prev = (all_preds[-4:end_pred_idx], self.all_trimap[-4:end_pred_idx], self.all_gts[-4:end_pred_idx])
        self._prev_len = None if prev is None else len(prev[0])
        end_all_idx = -2 if not is_last else len(all_preds)
        cur = (all_preds[-3:end_all_idx], self.all_trimap[-3:end_all_idx], self.all_gts[-3:end_all_idx])
        if len(all_preds) > 3:
            self.all_preds, self.all_gts = self.all_preds[-3:], self.all_gts[-3:]
            self.all_trimap, self.all_image_names = self.all_trimap[-3:], self.all_image_names[-3:]
        return {'save': save, 'current': cur, 'previous': prev}
