"""TEST INFRASTRUCTURE ONLY. numpy restatement of the window bookkeeping of eval_video, maggie/engine/test.py:237-286, statement by
statement (lists/arrays exactly as there; `prev_preds` on the last clip is read as "all stored predictions", the only reading under which the
reference's `len(prev_preds)` is defined on a first-and-last clip). Parity unpinned: eval_video needs a model and a data loader to run."""
import numpy as np


class Window:
    def __init__(self):
        self.all_preds, self.all_gts, self.all_trimap, self.all_image_names = [], [], [], []

    def push(self, alpha, alpha_gt, trimap, image_names, is_first, is_last):
        if is_first:
            self.all_preds, self.all_gts, self.all_trimap, self.all_image_names = alpha[0], alpha_gt[0], trimap[0], list(image_names)
        else:
            self.all_gts = np.concatenate([self.all_gts, alpha_gt[0, 2:]], axis=0)
            self.all_trimap = np.concatenate([self.all_trimap, trimap[0, 2:]], axis=0)
            self.all_image_names = self.all_image_names + list(image_names[2:])
            self.all_preds = np.concatenate([self.all_preds[:-1], alpha[0, 1:]], axis=0)
        all_preds = self.all_preds
        end_idx = 1 if not is_last else len(all_preds)
        save = (self.all_image_names[:end_idx], all_preds[None, :end_idx])
        end_pred_idx = -3 if not is_last else len(all_preds)
        prev = None
        if len(all_preds) > 3:
            prev = (all_preds[-4:end_pred_idx], self.all_trimap[-4:end_pred_idx], self.all_gts[-4:end_pred_idx])
        end_all_idx = -2 if not is_last else len(all_preds)
        cur = (all_preds[-3:end_all_idx], self.all_trimap[-3:end_all_idx], self.all_gts[-3:end_all_idx])
        if len(all_preds) > 3:
            self.all_preds, self.all_gts = self.all_preds[-3:], self.all_gts[-3:]
            self.all_trimap, self.all_image_names = self.all_trimap[-3:], self.all_image_names[-3:]
        return {'save': save, 'current': cur, 'previous': prev}
