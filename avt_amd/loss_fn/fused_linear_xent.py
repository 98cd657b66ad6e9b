"""``LinearCrossEntropy``: a classifier and its cross entropy as one module for callers that own both ends.

The reference keeps them apart -- ``torch.nn.Linear`` inside ``BaseModel`` (models/base_model.py:203-216), ``MultiDimCrossEntropy``
inside ``BasicLossAccuracy`` (func/train_eval_ops.py:27-44, loss_fn/multidim_xentropy.py:11-25).  The drop-in model keeps that module
boundary and still runs both as ONE autograd node during training: ``Basic`` hands the labels to ``BaseModel``, whose classifiers
call ``HipLinear.forward_with_loss`` (models/classifiers.py -> ``avt_linear_softmax_xent_fwd / _bwd``).  This class is the same node
with ``torch.nn.Linear``'s parameters (``weight (out, in)``, ``bias``) for code that is not built around ``BaseModel``."""
from ..models.classifiers import HipLinear


class LinearCrossEntropy(HipLinear):
    def __init__(self, in_features, out_features, bias=True, ignore_index=-1):
        super().__init__(in_features, out_features, bias=bias)
        self.weight.data.normal_(0, 0.01)
        self.ignore_index = ignore_index

    def forward(self, feats, target):
        """feats (..., in_features), target (...) int64 -> (loss (...), rank (...), logits (..., out_features)); loss is un-reduced,
        0 where target == ignore_index; rank = number of logits above the target's (-1 where ignored).  The logits are a
        differentiable output: a gradient that reaches them is added to the cross entropy's before the weight / bias / input
        gradients are formed."""
        logits, loss, rank = self.forward_with_loss(feats, target, self.ignore_index)
        return loss, rank, logits
