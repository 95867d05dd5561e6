"""Runs a tools/ script against ANOTHER build of the library (side-by-side comparisons on one GPU box):
OMR_AB_LIB=<path to a libomrdeskew.so variant> python3 tools/ab_lib.py tools/<script>.py [args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
from oics import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(os.environ["OMR_AB_LIB"])
sys.argv = [os.path.abspath(sys.argv[1])] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
