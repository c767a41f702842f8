"""Per-row phase stamps of the encoder forward (rk_enc_probe): when each row workgroup starts, how long it
waits for its row pointer + first entries, its gather and its epilogue; the five longest rows.
    python tools/probes/enc_phase_probe.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from recoder_amd import _lib, synthetic
from recoder_amd._lib import check, ptr
from recoder_amd.device import Block, DeviceCSR, current_stream
lib = _lib.load(); raw = lib
dev = torch.device("cuda"); B, h = 500, 200
csr = synthetic.ml20m_like(seed=0, n_users=20000); dcsr = DeviceCSR(csr); n_items = csr.shape[1]
f = dict(dtype=torch.float32, device=dev)
W = torch.randn(n_items, h, **f) * 0.05; bias = torch.zeros(h, **f)
users = torch.arange(1000, 1000 + B, dtype=torch.int64, device=dev)
blk = Block(B, int(np.sort(dcsr.degrees)[-B:].sum()), n_items, dev); blk.collate(dcsr, users)
Z = torch.empty(B, h, **f); st = current_stream()
fn = lambda: check(lib.rk_ae_encode_fwd(blk.ref, 0, B, ptr(W), ptr(bias), h, None, 0.5, 7, 3, ptr(users), 1, ptr(Z), st))
for _ in range(5): fn()
torch.cuda.synchronize()
probe = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
raw.rk_enc_probe(probe.data_ptr())
big = torch.empty(64 << 20, dtype=torch.float32, device=dev); big.fill_(1.0)   # flush caches
torch.cuda.synchronize()
fn(); torch.cuda.synchronize()
raw.rk_enc_probe(None)
a = probe.cpu().numpy().reshape(-1, 8)[:B]
t0 = a[:, 0].min(); rel = (a[:, :4] - t0) * 0.01
q = lambda x: "min %5.1f med %5.1f p90 %5.1f max %5.1f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
print("start     ", q(rel[:, 0])); print("idx+entry ", q(rel[:, 1] - rel[:, 0])); print("gather    ", q(rel[:, 2] - rel[:, 1])); print("epilogue  ", q(rel[:, 3] - rel[:, 2])); print("last end  %.1f us" % rel[:, 3].max())
deg = np.diff(blk.indptr.cpu().numpy())[:B]
heavy = np.argsort(-deg)[:5]
for r in heavy: print("row deg %d: start %.1f entry %.1f gather %.1f epi %.1f" % (deg[r], rel[r,0], rel[r,1]-rel[r,0], rel[r,2]-rel[r,1], rel[r,3]-rel[r,2]))
