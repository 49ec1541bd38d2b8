"""Composing existing autograd Functions into ONE autograd node.

A Python `autograd.Function` application costs the host 25-55 us per direction on this path whatever its kernels do
(profiles/r05_host_profile.txt: 127 applications, 6.9 ms cumulative per step), and the step is within 2 ms of host-bound.
Chains that always run together (sparse convolution -> BatchNorm -> ReLU; the halves of an encoder layer) are therefore
applied as one node whose forward / backward call the static forward / backward of the existing Functions with a
stand-in context -- the same kernels in the same order, no second implementation of any of them."""


class Ctx:
    """Stand-in for an autograd context: what the static forward / backward of this package's Functions use of one
    (save_for_backward / saved_tensors, attributes, needs_input_grad, mark_non_differentiable)."""

    def __init__(self, n_inputs=16):
        self.saved_tensors = ()
        self.needs_input_grad = (True,) * n_inputs

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


def pack(ctx, sub, prefix):
    """The stand-in's saved tensors (to go on the real context's save list: autograd must own them); its attributes are
    parked on the real context under `prefix`."""
    setattr(ctx, prefix + "_attrs", {k: v for k, v in sub.__dict__.items() if k not in ("saved_tensors", "needs_input_grad")})
    return list(sub.saved_tensors)


def unpack(ctx, tensors, prefix, needs_input_grad=None):
    sub = Ctx()
    sub.saved_tensors = tuple(tensors)
    sub.__dict__.update(getattr(ctx, prefix + "_attrs"))
    if needs_input_grad is not None:
        sub.needs_input_grad = tuple(needs_input_grad)
    return sub
