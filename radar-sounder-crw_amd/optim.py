"""The reference's optimizer -- ``torch.optim.Adam(model.parameters(), lr)`` with its defaults (scripts/train.py:54,69) -- as ONE
launch per step: the parameters become views of one flat fp32 buffer (like their gradients in ``dist.FlatGradBucket``) and
``crw_adam_step`` updates the whole buffer.  Same arithmetic as torch's default implementation, operation for operation
(``tests/test_hip_parity.py::test_flat_adam_matches_torch_adam``); ``state_dict`` keys / values of the module are unchanged (the
parameters keep their names and shapes, only their storage moves)."""
import torch

import crw_hip


class FlatAdam:
    """``FlatAdam(bucket, lr)``: bucket is the ``dist.FlatGradBucket`` that owns the flat gradient of the same parameters."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.bucket, self.lr, self.betas, self.eps = bucket, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        params = bucket.params
        if not params or any(p.dtype != torch.float32 or not p.is_cuda for p in params):
            raise ValueError("FlatAdam needs fp32 parameters on the GPU (the HIP path); use torch.optim.Adam elsewhere")
        self.flat = torch.cat([p.data.reshape(-1) for p in params])
        o = 0
        for p in params:  # the module's parameters become views of the flat buffer (same order as the flat gradient)
            p.data = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()
        self.m, self.v, self.t = torch.zeros_like(self.flat), torch.zeros_like(self.flat), 0

    def step(self):
        """after ``bucket.all_reduce_mean()``: the flat gradient is complete."""
        self.t += 1
        crw_hip.adam_step(self.flat, self.bucket.flat, self.m, self.v, self.t, self.lr, self.betas[0], self.betas[1], self.eps)
