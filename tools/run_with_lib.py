#!/usr/bin/env python
"""tools only: run a script against an ablation build of the library:  python tools/run_with_lib.py <lib.so> <script.py> [args...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ic_gan_amd._lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name="__main__")
