"""GPU job: the oracle's pin, on the GPU box. tests/test_oracle_vs_ref.py (C restatement == the UNMODIFIED reference build
oracle/_ref/libpolarc_ref.so: tables, decode_scl_llr, decode_scl_p1, whole get_bler_quick runs) normally runs in the build
container only; the prebuilt library travels with the snapshot, so the same checks run here too — against the GPU box's
own glibc / libm, the ones every `-m gpu` parity test's expected values are computed with. Skipped (not failed) where the
reference build did not travel."""
import pytest

import oracle_lib
import test_oracle_vs_ref as _pin

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref did not travel to this box")]

test_oracle_equals_reference_on_this_box = _pin.test_oracle_equals_reference
test_probability_domain_equals_reference_on_this_box = _pin.test_probability_domain_equals_reference
test_get_bler_quick_equals_reference_on_this_box = _pin.test_get_bler_quick_equals_reference
