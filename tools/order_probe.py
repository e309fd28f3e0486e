import sys
sys.path[:0] = ["/root/repo", "/root/repo/pde-solver_amd"]
import os
print("env HIP/ROCR:", {k: v for k, v in os.environ.items() if "VISIBLE" in k or "HSA" in k})
from pdegpu import _lib as L
ctx = L.context(0)
print("lib ctx ok")
import torch
try:
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    s = torch.cuda.Stream()
    print("stream ok", s)
except Exception as e:
    print("torch after lib FAILED:", repr(e)[:300])
