"""`spconv.pytorch` subset used by the EFG backbones, MI355X-native (see core.py)."""
from .core import (Rulebook, SiteIndex, SparseConv3d, SparseConvTensor, SparseModule, SparseSequential,  # noqa: F401
                   SubMConv3d, conv_bn_act, conv_pair_bn_act, is_spconv_module, run_modules, weights_updated)

pytorch = None  # `import efg_amd.spconv as spconv; spconv.SparseConvTensor` and `spconv.pytorch.*` both work


def _self_as_pytorch():
    import sys

    global pytorch
    pytorch = sys.modules[__name__]


_self_as_pytorch()
