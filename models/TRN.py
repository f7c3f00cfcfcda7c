"""``RelationModuleMultiScale`` -- API surface only.

The reference defines this multi-scale TRN (models/TRN.py:13-74) but nothing imports it (SURVEY fact 1:
dead code; the TRN on the executed path is TRNPooling).  The class is kept importable with the same
constructor so that ``from models.TRN import RelationModuleMultiScale`` keeps working; it has no HIP
implementation because it is never executed on the hot path."""
import torch


class RelationModuleMultiScale(torch.nn.Module):
    def __init__(self, img_feature_dim, num_bottleneck, num_frames, *args, **kwargs):
        super().__init__()
        self.img_feature_dim, self.num_bottleneck, self.num_frames = img_feature_dim, num_bottleneck, num_frames

    def forward(self, *args, **kwargs):
        raise NotImplementedError("RelationModuleMultiScale is dead code in the reference and is not on the "
                                  "MI355X hot path; use models.temporal_pooling.trn_pooling.TRNPooling")
