"""Build librecoder_hip.so (hand-written HIP kernels + C ABI) for gfx950.

    python -m recoder_amd.build [--force]

hipcc cross-compiles without a GPU; the built library stays in-tree
(recoder_amd/csrc/librecoder_hip.so, git-ignored) so that it travels with the
repository snapshot to the GPU box.
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "librecoder_hip.so")
SOURCES = ["capi.hip", "collate.hip", "encoder.hip", "gemm.hip", "decode16.hip", "linear.hip", "dw3.hip", "pgemm.hip", "fdecode.hip", "optim.hip", "topk.hip", "step.hip", "comm.hip"]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has a unified register
# file); without it hipcc copied all accumulators AGPR<->VGPR around every k-tile
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-fvisibility=hidden",       # (exports: what include/recoder_hip.h declares, nothing else)
         "--offload-compress"]        # (the gfx950 code objects zstd-compressed in the bundle: 3.6 -> ~1 MB)


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
  hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
  headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
      [os.path.join(os.path.dirname(CSRC), "..", "include", "recoder_hip.h"),
       os.path.join(os.path.dirname(CSRC), "..", "include", "recoder_hip_probe.h")]
  objs = []
  procs = []
  for src in SOURCES:
    s = os.path.join(CSRC, src)
    o = os.path.join(CSRC, src.replace(".hip", ".o"))
    objs.append(o)
    if force or _stale(o, [s] + headers):
      cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
      if verbose:
        print(" ".join(cmd), flush=True)
      procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
  failed = False
  for src, p in procs:
    out, _ = p.communicate()
    if out and verbose:
      sys.stdout.write(out.decode(errors="replace"))
    if p.returncode != 0:
      failed = True
      print("FAILED:", src)
  if failed:
    raise RuntimeError("hipcc failed")
  if force or procs or _stale(LIB, objs):
    # (-z defs: an internal helper that is declared but defined nowhere must fail HERE, not at dlopen on the GPU box)
    cmd = [hipcc, "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-Wl,-z,defs", "-o", LIB] + objs + ["-ldl"]
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
  return LIB


if __name__ == "__main__":
  build_library(force="--force" in sys.argv)
  print("built", LIB)
