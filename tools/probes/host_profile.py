"""cProfile of the per-entry training path (C4-shaped MF) -- where the host time of a step goes."""
import cProfile, pstats, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs as bc
from recoder_amd.data import RecommendationDataset
from recoder_amd.model import Recoder
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
FULL = len(sys.argv) > 2 and sys.argv[2] == "full"      # whole epochs (graph replay where eligible)
c = bc.CONFIGS[name]
csr = c["data"]()
torch.manual_seed(0)
rec = Recoder(model=c["model"](), use_cuda=True, optimizer_type="adam", loss=c["loss"])
ds = RecommendationDataset(csr)
kw = dict(batch_size=500, lr=1e-3, weight_decay=c["wd"], negative_sampling=True)
rec.train(ds, num_epochs=1, **kw) if FULL else rec.train(ds, num_epochs=1, iters_per_epoch=20, **kw)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
rec.train(ds, num_epochs=2, **kw) if FULL else rec.train(ds, num_epochs=rec.current_epoch, iters_per_epoch=150, **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
