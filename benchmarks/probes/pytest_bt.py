#!/usr/bin/env python3
"""pytest with a native backtrace on SIGSEGV (segv_backtrace.cpp): pytest_bt.py <pytest args>"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ctypes.CDLL(os.path.join(ROOT, "benchmarks", "probes", "libsegvbt.so")).install_segv_backtrace()
import pytest

sys.exit(pytest.main(["-p", "no:faulthandler"] + sys.argv[1:]))
