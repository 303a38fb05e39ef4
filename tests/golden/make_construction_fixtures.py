#!/usr/bin/env python3
"""tests/golden/make_construction_fixtures.py — the Monte-Carlo construction tables the reference SHIPS for the BICM receiver
(PolarM/CodeConstructionData/MC_block_length_1024_512_*_bicm*.txt: per-channel error counts of the genie-aided SC decoder,
written by PolarCode.m:120-124) as one small data fixture, tests/golden/construction_tables.npz. Data only (integers); run HERE
(the reference tree does not travel to the GPU box). The four `*_mlc*` tables belong to the multi-level-coding receiver, which is
out of scope and are not copied.

Run counts: the name carries them (`_250000`) or, for the older file names without one, the reference's default num_runs = 100e3
(PolarCode.m:96-98; their largest count is 0.5 x 100 000 within sampling noise)."""
import glob
import os
import re

import numpy as np

SRC = "/root/reference/PolarM/CodeConstructionData"
HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
names = []
for f in sorted(glob.glob(os.path.join(SRC, "MC_block_length_1024_512_*_bicm*.txt"))):
    m = re.match(r"MC_block_length_1024_512_cc_method_monte-carlo_cc_param_(-?[\d.]+)_([a-z0-9-]+)_bicm(?:_(\d+))?\.txt", os.path.basename(f))
    snr, const, runs = float(m.group(1)), m.group(2), int(m.group(3) or 100000)
    key = f"{const}_{m.group(1)}_{runs}"
    counts = np.loadtxt(f).astype(np.int64)
    assert counts.size == 1024
    out[key + "/counts"] = counts.astype(np.int32)
    out[key + "/meta"] = np.array([snr, runs], np.float64)
    names.append((key, os.path.basename(f)))
out["keys"] = np.array([k for k, _ in names])
out["files"] = np.array([f for _, f in names])
np.savez_compressed(os.path.join(HERE, "construction_tables.npz"), **out)
print(len(names), "tables:", [k for k, _ in names])
