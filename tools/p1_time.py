#!/usr/bin/env python3
"""tools/p1_time.py — the probability-domain members through their host entry points at batch 8192 (bench.py p1_record), with the
first codewords checked against the CPU restatement."""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, polar_amd
a = argparse.Namespace(n=11, K=1024, crc=16, L=32, ebno=2.0, seed=2024)
C.CDLL(None).srand(C.c_uint(1))
code = polar_amd.PolarCode(a.n, a.K, 0.32, a.crc)
print(json.dumps(bench.p1_record(a, code), indent=1))
