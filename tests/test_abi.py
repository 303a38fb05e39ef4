"""CPU suite: the C-ABI library loads, exports every symbol include/polar_amd.h declares, builds
the reference's tables on the host, validates arguments, and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libc = C.CDLL(None)


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(polar_[a-z0-9_]+)\s*\(", hdr)))


def test_exports_every_declared_symbol(built_lib):
    """Both libraries (the product and the test build) export every entry point of include/polar_amd.h and of
    include/polar_amd_debug.h, and nothing but `polar_*` C symbols (kernel launchers and internals stay local)."""
    import subprocess
    from polar_amd import build
    api, dbg = _declared("polar_amd.h"), _declared("polar_amd_debug.h")
    assert len(api) >= 25 and "polar_debug_set" in dbg and not [n for n in api if n.startswith("polar_debug")]
    assert "polar_amd_debug.h" not in re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "polar_amd.h")).read(), flags=re.S)
    for path in (built_lib, build.build_test()):
        lib = C.CDLL(path)
        missing = [n for n in api + dbg if not hasattr(lib, n)]
        assert not missing, (path, missing)
        exported = [l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", path], text=True).splitlines() if " T " in l]
        foreign = [n for n in exported if not n.startswith("polar_")]
        assert not foreign, (path, foreign[:5])


def test_fault_injection_exists_only_in_the_test_build(built_lib):
    """The product library has no key that makes a sweep fail, stall or share a device (include/polar_amd_debug.h): polar_debug_set
    rejects them as unknown; the test build (-DPOLAR_TEST_HOOKS) accepts them. The measurement knobs exist in both."""
    import polar_amd
    from polar_amd import build
    hooks = ("share_device", "fail_device", "fail_collective", "force_workers", "stall_device", "stall_ms")
    knobs = ("mode_override", "lat_max_b", "host_lanes", "host_prefault", "no_rccl", "multi_timeout_s")
    g = polar_amd.PolarCode(5, 16, 0.32, 0)
    assert g.debug_get("test_hooks") == 0
    for k in hooks:
        with pytest.raises(polar_amd.PolarError, match="unknown key"):
            g.debug_set(k, 1)
    for k in knobs:
        g.debug_set(k, 0)
    try:
        polar_amd.use_library(build.build_test())
        t = polar_amd.PolarCode(5, 16, 0.32, 0)
        assert t.debug_get("test_hooks") == 1
        for k in hooks + knobs:
            t.debug_set(k, 0)
        t.close()
    finally:
        polar_amd.use_library(None)
    src = open(os.path.join(ROOT, "polar_amd", "csrc", "polar_debug.cpp")).read()
    for k in hooks:          # every fault-injection key sits inside a POLAR_TEST_HOOKS block
        i = src.index('"%s"' % k)
        assert src.rfind("#ifdef POLAR_TEST_HOOKS", 0, i) > src.rfind("#endif", 0, i), k


def test_no_oracle_or_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "polar_amd.h")).read()
    assert "torch" not in hdr and "at::" not in hdr and "oracle" not in hdr.lower().replace("test oracle", "")


@pytest.mark.parametrize("name", G.code_names())
def test_host_construction_matches_reference_tables(built_lib, name):
    """polar_create = PolarCode ctor: same frozen set, same info order (std::sort tie order),
    same rand() CRC matrix as the golden tables captured from the reference."""
    import polar_amd
    c, frozen, order, crcm = G.tables(name)
    libc.srand(1)
    if c.get("explicit_tables"):
        g = polar_amd.PolarCode.from_tables(c["n"], c["K"], c["crc"], frozen, order, None)
    else:
        g = polar_amd.PolarCode(c["n"], c["K"], c["eps"], c["crc"])
    assert (g.frozen_bits == frozen).all()
    assert (g.channel_order_descending == order).all()
    assert (g.crc_matrix == crcm).all()
    br = g.bit_rev_order
    n = c["n"]
    assert all(int(br[i]) == int(format(i, f"0{n}b")[::-1], 2) for i in range(1 << n))
    # explicit-table constructor round trip
    g2 = polar_amd.PolarCode.from_tables(c["n"], c["K"], c["crc"], frozen, order, crcm if c["crc"] else None)
    assert (g2.frozen_bits == frozen).all() and (g2.crc_matrix == crcm).all()


def test_argument_validation(built_lib):
    import polar_amd
    with pytest.raises(polar_amd.PolarError):
        polar_amd.PolarCode(0, 1, 0.32, 0)
    with pytest.raises(polar_amd.PolarError):
        polar_amd.PolarCode(16, 1024, 0.32, 0)          # reference: uint16_t block length
    with pytest.raises(polar_amd.PolarError):
        polar_amd.PolarCode(5, 30, 0.32, 8)             # K + crc > N
    g = polar_amd.PolarCode(5, 16, 0.32, 4)
    fr = g.frozen_bits.copy()
    fr[0] ^= 1
    with pytest.raises(polar_amd.PolarError):
        polar_amd.PolarCode.from_tables(5, 16, 4, fr, g.channel_order_descending, g.crc_matrix)
    for badL in (0, 65, -3):
        with pytest.raises(polar_amd.PolarError):
            g.decode_scl_llr(np.zeros(32), badL)
    assert abs(g.snr_sqrt_linear(2.0) - 10 ** 0.1 * np.sqrt(0.5)) < 1e-12
    # Monte-Carlo construction entry point: arguments are checked before any device work
    for bad in (dict(n=0), dict(n=16), dict(cons=9), dict(cons="qam16"), dict(runs=-1)):
        with pytest.raises(polar_amd.PolarError):
            polar_amd.mc_construction(bad.get("n", 5), 1.0, bad.get("runs", 10), bad.get("cons", "bpsk"))
    assert (polar_amd.mc_construction(5, 1.0, 0, "bpsk") == 0).all()          # zero runs: nothing to do, no device needed
    with pytest.raises(polar_amd.PolarError):
        g.decode_scl_llr(np.zeros(32, np.float32), 0)


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback(built_lib):
    import polar_amd
    g = polar_amd.PolarCode(5, 16, 0.32, 4)
    with pytest.raises(polar_amd.PolarError, match="no HIP device|no CPU"):
        g.decode_scl_llr(np.zeros(32), 4)
    with pytest.raises(polar_amd.PolarError):
        g.encode(np.zeros(16, np.uint8))
    with pytest.raises(polar_amd.PolarError, match="no HIP device|no CPU"):
        polar_amd.mc_construction(5, 1.0, 10, "bpsk")
    with pytest.raises(polar_amd.PolarError, match="no HIP device|no CPU"):
        g.decode_scl_llr(np.zeros(32, np.float32), 4)


def test_product_does_not_reference_oracle():
    """The product path must not import/link the oracle (it is test infrastructure)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "polar_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".m")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle_lib" not in txt and "polar_oracle" not in txt, f


def test_construction_file_loader(built_lib, tmp_path):
    """PolarM Monte-Carlo construction data (PolarCode.m:111-135): the 16-ASK file's counts are kept
    as a fixture; loading them from a text file must give the fixture's frozen set and info order."""
    import polar_amd
    z, m = G.load()
    counts = z["cfg5_n10_k512_ask16/counts"]
    f = tmp_path / "MC_block_length_1024_512_cc.txt"
    f.write_text("".join(f"{int(c)} \n" for c in counts))       # '%d \n' as PolarCode.m:122
    g = polar_amd.PolarCode.from_construction_file(str(f), 512)
    c, frozen, order, crcm = G.tables("cfg5_n10_k512_ask16")
    assert (g.frozen_bits == frozen).all() and (g.channel_order_descending == order).all()


def test_mex_gateway_layout_helpers(tmp_path):
    """The MEX gateway's layout helpers on their own: the column-major <-> codeword-contiguous conversions of a batch (blocked,
    multi-threaded) and the N x B / B x N / vector / square layout rule, against a naive loop (tests/mex_runtime/layout_test.cpp).
    The gateway itself is compiled, linked and RUN by tests/test_mex_gateway.py."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "layout_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(root, "polar_amd", "matlab"),
                           os.path.join(root, "tests", "mex_runtime", "layout_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_weak_unfrozen_leaves_are_classified_at_creation(built_lib):
    """Host side of the round-3 parity guard (HISTORY.md "Where bit-exactness ends"): codes a construction produces for an
    ordinary channel have no unfrozen leaf in the worst synthetic channels; rates near 1 / a design parameter that does not
    describe the channel do; the reference's 16-ASK BICM table has one (its bit levels are unequal: under a BEC it looks
    weak, on its own channel it is not — the device guard looks at the VALUE before it acts)."""
    import polar_amd
    L = polar_amd.lib()
    L.polar_get_weak_leaves.restype = C.c_int
    L.polar_get_weak_leaves.argtypes = [C.c_void_p]
    def weak(g):
        return L.polar_get_weak_leaves(g._h)
    for (n, K, eps, crc) in [(11, 1024, 0.32, 16), (11, 1024, 0.32, 0), (9, 256, 0.32, 0), (10, 512, 0.32, 0), (10, 614, 0.5, 0), (4, 10, 0.32, 0)]:
        assert weak(polar_amd.PolarCode(n, K, eps, crc)) == 0, (n, K, eps, crc)
    assert weak(polar_amd.PolarCode(9, 505, 0.7, 0)) > 100
    assert weak(polar_amd.PolarCode(10, 1022, 0.32, 0)) > 300
    counts = G.load()[0]["cfg5_n10_k512_ask16/counts"]
    assert weak(polar_amd.PolarCode.from_counts(counts, 512)) == 1


def test_explicit_tables_with_weak_leaves_return_a_status_not_an_error(built_lib):
    """polar_create_explicit accepts a table that leaves unfrozen leaves in the worst synthetic channels (the reference does
    too) but says so: POLAR_W_WEAK_LEAVES, a positive non-error status with a valid handle (the Python mirror turns it into a
    PolarWeakLeavesWarning); ordinary tables return POLAR_OK."""
    import warnings
    import polar_amd
    L = polar_amd.lib()
    src = polar_amd.PolarCode(9, 505, 0.7, 0)                     # K = 505 of 512 at eps 0.7: > 100 weak leaves
    fr, od = src.frozen_bits, src.channel_order_descending
    h = C.c_void_p()
    rc = L.polar_create_explicit(C.c_int(9), C.c_int(505), C.c_int(0), fr.ctypes.data_as(C.POINTER(C.c_uint8)),
                                 od.ctypes.data_as(C.POINTER(C.c_uint16)), None, C.byref(h))
    assert rc == polar_amd.POLAR_W_WEAK_LEAVES == 1 and h.value
    assert b"unfrozen leaves" in L.polar_last_error()
    L.polar_get_weak_leaves.restype = C.c_int
    assert L.polar_get_weak_leaves(h) == L.polar_debug_weak_leaves(h) == src.weak_leaves > 100
    L.polar_destroy(h)
    with pytest.warns(polar_amd.PolarWeakLeavesWarning):
        g = polar_amd.PolarCode.from_tables(9, 505, 0, fr, od)
    assert g.weak_leaves == src.weak_leaves
    ok = polar_amd.PolarCode(9, 256, 0.32, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        g2 = polar_amd.PolarCode.from_tables(9, 256, 0, ok.frozen_bits, ok.channel_order_descending)
    assert g2.weak_leaves == 0


def test_environment_knobs_are_read_once_and_validated(built_lib, monkeypatch):
    """POLAR_MODE & co. are measurement knobs: read when a handle is created (never inside a decode), and validated — an
    out-of-range value fails the creation instead of silently selecting a kernel path. polar_debug_set validates too, and the
    fault-injection hooks have no environment form."""
    import polar_amd
    monkeypatch.setenv("POLAR_MODE", "7")
    with pytest.raises(polar_amd.PolarError, match="POLAR_MODE"):
        polar_amd.PolarCode(5, 16, 0.32, 0)
    monkeypatch.setenv("POLAR_MODE", "2")
    g = polar_amd.PolarCode(5, 16, 0.32, 0)
    assert g.debug_get("mode_override") == 2
    monkeypatch.setenv("POLAR_MODE", "1")                          # (appears later: the existing handle keeps what it read)
    assert g.debug_get("mode_override") == 2
    monkeypatch.delenv("POLAR_MODE")
    g2 = polar_amd.PolarCode(5, 16, 0.32, 0)
    assert g2.debug_get("mode_override") == -1
    with pytest.raises(polar_amd.PolarError):
        g2.debug_set("mode_override", 3)
    with pytest.raises(polar_amd.PolarError, match="unknown key"):
        g2.debug_set("no_such_knob", 1)
    for k in ("force_rccl", "no_rccl", "sc_no_fold", "no_tables"):
        g2.debug_set(k, 1)
    csrc = os.path.join(ROOT, "polar_amd", "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith((".cpp", ".h", ".hip")))
    body = src[src.index("int read_env_knobs("):]
    body = body[:body.index("\n}\n")]
    assert src.count("getenv(") == body.count("getenv(") > 0      # every getenv of the library sits in read_env_knobs
    assert "FAIL" not in body and "SHARE" not in body              # (no environment form of the test hooks)


def test_build_refuses_development_macros_under_the_product_name():
    """POLAR_DEFS selects instrumented builds that measure instead of decoding: polar_amd/build.py accepts only names it knows,
    and only together with POLAR_BUILD_TAG (a library name of its own) — a typo or a stray variable cannot ship a decoder that is
    silently not bit-exact as libpolar_amd.so."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("POLAR_BUILD_TAG", None)
    for defs, msg in (("POLAR_MARGINN", "unknown macro"), ("POLAR_MARGIN", "without POLAR_BUILD_TAG"), ("POLAR_EXPERIMENT_FAST_DIV", "unknown macro")):
        r = subprocess.run([sys.executable, "-m", "polar_amd.build"], cwd=ROOT, env=dict(env, POLAR_DEFS=defs), capture_output=True, text=True)
        assert r.returncode != 0 and msg in (r.stderr + r.stdout), (defs, r.stderr[-300:])
    src = "".join(open(os.path.join(ROOT, "polar_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "polar_amd", "csrc")))
    assert "POLAR_EXPERIMENT" not in src
