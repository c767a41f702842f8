"""bench.py's contract with the driver: ONE JSON line, last on stdout, with the agreed keys; also for
the multi-rank code path (two processes on one GPU over gloo -- marked INVALID, a code-path test)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int,
        "ms_per_step": float, "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
        "config": dict, "roofline": dict}


def check_line(out, n_gpus, steps, warmup):
  lines = [l for l in out.strip().splitlines() if l.strip()]
  d = json.loads(lines[-1])                        # the LAST line of stdout
  for k, t in KEYS.items():
    assert k in d and isinstance(d[k], t), (k, d.get(k))
  assert "vs_baseline" in d and d["vs_baseline"] is None
  assert d["metric"] == "train_users_per_sec" and d["unit"] == "users/s"
  assert (d["n_gpus"], d["steps"], d["warmup"]) == (n_gpus, steps, warmup)
  assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
  assert "workload" in d["config"] and "model" not in d["config"]
  assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
  r = d["roofline"]
  for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernels", "step"):
    assert k in r, k
  assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
  # (the lazy dense Adam delivers SURVEY 8(d)'s dense-sweep bytes while moving fewer: its own bytes are priced in
  # roofline.lazy_sweep, whose fraction is the one bounded by 1)
  assert 0.0 < r["frac"] and len(r["kernels"]) >= 5
  own = r.get("lazy_sweep")
  assert (own is None and r["frac"] < 1.0) or (0.0 < own["frac"] < 1.0 and own["own_bytes"] < r["achieved"] * 1e9 * r["avg_launch_ms"] * 1e-3 * 1.001)
  return d


def test_single_gpu_line():
  r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "8", "--warmup", "4", "--cpu-seconds", "2"],
                     cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  d = check_line(r.stdout, 1, 8, 4)
  c = d["cpu_baseline"]
  assert c["kind"] == "port" and c["cores"] >= 1 and c["host_cores"] >= c["cores"] and c["value"] > 0
  assert d["config"]["api"] == "Recoder.train" and d["config"]["graph_replay"] is True
  # Recall@20 of the trained state next to the throughput it qualifies: product == oracle to 4 decimals
  assert d["recall_match_4dp"] is True and 0.0 <= d["recall_at_20"] <= 1.0, d.get("recall")
  assert abs(d["recall"]["value"] - d["recall"]["oracle"]) < 5e-5
  assert d["roofline"]["traffic"] is None or "NOT collected in this run" in d["roofline"]["traffic_source"]


@pytest.mark.parametrize("abandon_alt", [False, True])
def test_two_rank_code_path_on_one_gpu(abandon_alt):
  """abandon_alt: the extra item-parallel run behind the graded one is cut off by its watchdog
  (as if it hung): the graded line must still come out, last on stdout, and every rank exit 0."""
  env = dict(os.environ, MASTER_ADDR="127.0.0.1")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", "29598" if abandon_alt else "29597", "bench.py",
         "--gpus", "2", "--steps", "6", "--warmup", "3", "--one-gpu-gloo"]
  if abandon_alt:
    cmd += ["--alt-timeout", "0.05"]
  r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
  assert r.returncode == 0, r.stderr[-3000:]
  d = check_line(r.stdout, 2, 6, 3)
  assert d["config"]["global_batch"] == 2 * d["config"]["batch_size_per_gpu"]
  assert "INVALID" in d and d["config"]["parallelism"].startswith("dp2")
  alt = d["config"]["alt_item_parallel"]
  if abandon_alt:
    assert alt and "abandoned" in alt["error"]
  else:
    assert alt and "error" not in alt and alt["value"] > 0 and alt["parallelism"].startswith("ip2")
