"""Rounding error of one narrow sparse-conv layer against the double-accumulating oracle (forward and dgrad), on the kernel the
library picks: run once as is and once with EFG_CONV_SMALL=0 (the switch is read once per process).  GPU box.
    python scripts/ubench/conv_small_error.py [cin cout]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from test_oracle_spconv import random_sparse  # noqa: E402

import efg_amd.spconv as spconv  # noqa: E402

oracle.build()
dev = torch.device("cuda:0")
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(5, 16), (16, 16), (16, 32), (32, 16)]
for cin, cout in shapes:
    rng = np.random.default_rng(cin + cout)
    batch, shape = 2, (9, 40, 40)
    idx, feat = random_sparse(rng, batch, shape, 12000, cin)
    torch.manual_seed(0)
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k").to(dev)
    x = spconv.SparseConvTensor(torch.from_numpy(feat).to(dev), torch.from_numpy(idx).to(dev), shape, batch)
    x.features.requires_grad_(True)
    y = conv(x)
    w = conv.weight.detach().cpu().numpy().reshape(cout, 27, cin)
    nbr = oracle.spconv_rulebook(idx, idx, batch, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    ref = oracle.spconv_forward(feat, w, None, nbr)
    go = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(torch.from_numpy(go).to(dev))
    dref = oracle.spconv_dgrad(go, w, nbr, feat.shape[0])
    ef = np.abs(y.features.detach().cpu().numpy() - ref).max() / np.abs(ref).max()
    ed = np.abs(x.features.grad.cpu().numpy() - dref).max() / np.abs(dref).max()
    print("EFG_CONV_SMALL=%s %2d -> %2d  rows %d  forward max error %.2e of the max, dgrad %.2e" % (
        os.environ.get("EFG_CONV_SMALL", "1"), cin, cout, feat.shape[0], ef, ed))
