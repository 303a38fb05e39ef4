"""CPU suite, build container only: the oracle against the live UNMODIFIED reference build
(oracle/_ref/libpolarc_ref.so). Skipped where the reference build did not travel."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from oracle_lib import Oracle

pytestmark = pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built here")
libc = C.CDLL(None)


@pytest.mark.parametrize("n,K,crc", [(6, 32, 0), (8, 128, 4), (9, 256, 0), (10, 512, 8), (11, 1024, 16)])
def test_oracle_equals_reference(oracle_built, n, K, crc):
    libc.srand(1)
    r = oracle_lib.Reference(n, K, 0.32, crc)
    libc.srand(1)
    o = Oracle(n, K, 0.32, crc)
    assert (r.frozen() == o.frozen()).all() and (r.order() == o.order()).all()
    assert (r.crc_matrix() == o.crc_matrix()).all() and (r.bitrev() == o.bitrev()).all()
    B = 24 if n >= 10 else 64
    for ebno in (0.5, 2.0):
        llr, _ = o.synth_llr(99, 10, B, o.snr_sqrt_linear(ebno))
        for L in (1, 3, 4, 32):
            assert (r.decode_scl_llr(llr, L) == o.decode_scl_llr(llr, L)).all()


def test_probability_domain_equals_reference(oracle_built):
    libc.srand(1)
    r = oracle_lib.Reference(8, 128, 0.32, 4)
    libc.srand(1)
    o = Oracle(8, 128, 0.32, 4)
    llr, _ = o.synth_llr(5, 0, 32, o.snr_sqrt_linear(2.0))
    for i in range(32):
        p1 = 1 / (1 + np.exp(llr[i]))
        for L in (1, 4, 8):
            assert (r.decode_scl_p1(p1, 1 - p1, L) == o.decode_scl_p1(p1, 1 - p1, L)).all()


def test_get_bler_quick_equals_reference(oracle_built):
    libc.srand(1)
    r = oracle_lib.Reference(8, 128, 0.32, 0)
    a = r.get_bler_quick([1.0, 2.0, 3.0], [1, 4])
    libc.srand(1)
    o = Oracle(8, 128, 0.32, 0)
    b = o.get_bler_quick_ref([1.0, 2.0, 3.0], [1, 4])
    assert (a == b).all()
