#!/usr/bin/env python
"""Per-kernel resource usage of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kres.py recoder_amd/csrc/decode16.hip"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def main():
  src = sys.argv[1]
  out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
  cur = None
  rows = {}
  for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
      cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
      rows[cur] = {}
      continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", l)
    if m and cur:
      rows[cur][m.group(1).strip()] = int(m.group(2))
  for k, v in rows.items():
    print("%-90s VGPR %3d AGPR %3d SGPR %3d scratch %4d occ %d LDS %6d" % (
        k[:90], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1), v.get("ScratchSize", -1),
        v.get("Occupancy", -1), v.get("LDS Size", -1)))
  if "error" in out:
    print(out[-3000:])


if __name__ == "__main__":
  main()
