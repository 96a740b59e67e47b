"""Does loading libivs.so before torch hide the GPU from torch?  (run on the GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "iv_interpolation_amd", "libivs.so")
a = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({lib!r}); import torch; print('lib first ->', torch.cuda.is_available())"], capture_output=True, text=True)
b = subprocess.run([sys.executable, "-c", f"import torch, ctypes; ctypes.CDLL({lib!r}); print('torch first ->', torch.cuda.is_available())"], capture_output=True, text=True)
c = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import pandas; from iv_interpolation_amd import engine; t = engine.require_device(); print('package, no explicit torch import ->', t.cuda.is_available())" % ROOT], capture_output=True, text=True)
for r in (a, b, c):
    print(r.stdout.strip(), r.stderr.strip()[-300:])
