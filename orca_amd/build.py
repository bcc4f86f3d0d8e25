"""Builds liborca_hip.so (the C-ABI library declared in include/orca_hip.h) in-tree with hipcc for
gfx950.  Run here on the CPU container (hipcc cross-compiles); the .so travels to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liborca_hip.so")
SOURCES = ["csr.hip", "tgat.hip", "tgat_bwd.hip", "gru_bwd.hip", "gru.hip", "state.hip", "plan.hip", "plan_dev.hip", "engine.hip", "optim.hip", "decoder.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++20", "-fPIC", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-result", "-Wno-unused-function"]


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
  hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
  # every header is a dependency of every object: the sampler and the split GRU live in headers
  inc = os.path.join(HERE, "..", "include")
  headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
            sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))
  objs, jobs = [], []
  os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
  for s in SOURCES:
    src = os.path.join(CSRC, s)
    obj = os.path.join(CSRC, "build", s.replace(".hip", ".o"))
    objs.append(obj)
    if force or _stale(obj, [src] + headers):
      jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

  def run(cmd):
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)

  with ThreadPoolExecutor(max_workers=4) as ex:
    list(ex.map(run, jobs))
  if jobs or _stale(OUT, objs):
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
  return OUT


if __name__ == "__main__":
  print(build(force="--force" in sys.argv, verbose=True))
