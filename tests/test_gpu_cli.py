"""GPU suite: the C++ host mirror (polar_amd/cpp/PolarCode.hpp) + CLI driver — the counterpart of
the reference's main.cpp — reproduces the reference's BLER table STATISTICALLY (the serial RNGs of
the reference cannot be reproduced by a parallel device), and the published curve shape
(results/polar_performance.jpeg anchor points)."""
import subprocess

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


def _run_cli(args):
    from polar_amd import build
    exe = build.build_cli()
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    rows = [[float(x) for x in l.split()] for l in out.stdout.strip().splitlines()]
    return np.array(rows)


def test_cli_reproduces_reference_table_statistically(built_lib):
    z, m = G.load()
    want = z["main_cpp/bler"]              # [L][ebno] from the unmodified reference (1000 runs, max_err 100)
    tab = _run_cli(["--runs", "4000", "--max-err", "1000000", "--batch", "4000"])
    assert tab.shape == (5, 6)
    assert np.allclose(tab[:, 0], [1.0, 1.25, 1.5, 1.75, 2.0])
    got = tab[:, 1:].T                      # -> [L][ebno]
    # binomial consistency: the reference estimate p_ref comes from >= ~140 runs (early stop) or 1000
    # runs; ours from 4000. Allow 4.5 sigma of the combined standard error (+ the small bias of
    # the reference's skip hack, which can only lower its estimate).
    n_ref = np.where(want > 0.101, 100.0 / np.maximum(want, 1e-9), 1000.0)
    se = np.sqrt(want * (1 - want) / n_ref + got * (1 - got) / 4000.0) + 1e-3
    assert (np.abs(got - want) < 4.5 * se + 0.01).all(), (got, want)
    # monotone in Eb/N0 and (weakly) in list size
    assert (np.diff(got, axis=1) <= 0.01).all()
    assert (np.diff(got, axis=0) <= 0.01).all()


def test_cli_crc_aided_l32_curve_shape(built_lib):
    """L=32 + 16-bit CRC: the published curve has BLER ~0.2 @1.0 dB, ~8e-3 @1.5 dB."""
    tab = _run_cli(["--crc", "16", "--L", "32", "--runs", "3000", "--max-err", "1000000", "--emin", "1.0",
                    "--emax", "1.51", "--estep", "0.5"])
    assert 0.08 < tab[0, 1] < 0.35, tab
    assert 0.001 < tab[1, 1] < 0.03, tab
