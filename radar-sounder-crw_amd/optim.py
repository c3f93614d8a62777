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

    def _check_views(self):
        """the parameters must still be the views of ``self.flat`` made at construction: ``module.to()/.float()/.cuda()`` or
        ``load_state_dict(assign=True)`` re-bind ``p.data`` and the step would then update a dead buffer without an error"""
        params = self.bucket.params
        first, last = params[0], params[-1]
        end = self.flat.data_ptr() + 4 * (self.flat.numel() - last.numel())
        if first.data_ptr() != self.flat.data_ptr() or last.data_ptr() != end or sum(p.numel() for p in params) != self.flat.numel():
            raise RuntimeError("FlatAdam: the module's parameters no longer alias the optimizer's flat buffer (moved / re-assigned "
                               "after FlatAdam was built?) -- rebuild the FlatGradBucket and FlatAdam")

    def step(self):
        """after ``bucket.all_reduce_mean()``: the flat gradient is complete."""
        self._check_views()
        self.t += 1
        crw_hip.adam_step(self.flat, self.bucket.flat, self.m, self.v, self.t, self.lr, self.betas[0], self.betas[1], self.eps)

    def state_dict(self):
        """moments, step count and hyper-parameters (checkpoint / resume, like ``torch.optim.Adam.state_dict``)"""
        return {"m": self.m.clone(), "v": self.v.clone(), "t": self.t, "lr": self.lr, "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        if sd["m"].numel() != self.flat.numel():
            raise ValueError("FlatAdam.load_state_dict: moment buffers of another parameter set")
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.t, self.lr, self.betas, self.eps = int(sd["t"]), float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"])
