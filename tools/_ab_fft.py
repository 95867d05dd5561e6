import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import os, sys, runpy
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
from oics import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), %r)
sys.argv = ["bench_fft.py", "64", "4", "both"]
runpy.run_path(os.path.join(ROOT, "tools", "bench_fft.py"), run_name="__main__")
'''
for rep in range(3):
    for lib in ("libomrdeskew.so", "libomrdeskew_prev.so"):
        out = subprocess.run([sys.executable, "-c", code % (ROOT, lib)], capture_output=True, text=True)
        print(lib, re.findall(r'"scans_per_s": [0-9.]*', out.stdout), out.stderr[-200:] if out.returncode else "")
