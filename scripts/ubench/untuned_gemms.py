import os, sys, torch
os.environ["PYTORCH_TUNABLEOP_UNTUNED_FILENAME"] = "gpurun_out/untuned.csv"
sys.path.insert(0, '.')
import torch.cuda.tunable as tunable
from efg_amd.engine import Trainer, synthetic_batch
dev = torch.device('cuda:0')
tr = Trainer(device=dev, seed=0)
tunable.record_untuned_enable(True)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(3): tr.step(pool[s % 2])
torch.cuda.synchronize()
