import sys, time, torch
sys.path.insert(0, '.')
from efg_amd.engine import Trainer, synthetic_batch
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = (sys.argv[1] == '1')
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(14):
    torch.cuda.synchronize(); t = time.perf_counter()
    tr.step(pool[s % 2])
    torch.cuda.synchronize(); print("step", s, "%.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
