"""ctypes bindings for the TEST-ONLY libraries under oracle/ (the CPU restatement and, when
present, the unmodified reference build oracle/_ref/).  Imported by tests/, smoke() and bench.py's
cpu_baseline leg only — never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libpolarc_ref.so")

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u64p = C.POINTER(C.c_uint64)


def _p(a, t):
    return a.ctypes.data_as(t)


def usable_cpus():
    """Host cores this process may actually use: the cgroup CPU quota when there is one (the GPU boxes report 256
    logical CPUs but run the container on a 16-CPU quota), else the affinity mask / CPU count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


class _Base:
    prefix = ""

    def __init__(self, lib, n, K, eps, crc):
        self.lib = lib
        self.n, self.N, self.K, self.crc = n, 1 << n, K, crc
        f = getattr(lib, self.prefix + "create")
        f.restype = C.c_void_p
        f.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        self.h = C.c_void_p(f(n, K, eps, crc))

    def _call(self, name, *args, restype=None):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        return f(self.h, *args)

    def close(self):
        if self.h:
            self._call("destroy")
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # tables
    def frozen(self):
        a = np.zeros(self.N, np.uint8)
        self._call("get_frozen", _p(a, _u8p))
        return a

    def order(self):
        a = np.zeros(self.N, np.uint16)
        self._call("get_order", _p(a, _u16p))
        return a

    def bitrev(self):
        a = np.zeros(self.N, np.uint16)
        self._call("get_bitrev", _p(a, _u16p))
        return a

    def crc_matrix(self):
        a = np.zeros((max(self.crc, 0), self.K), np.uint8)
        if self.crc:
            self._call("get_crc_matrix", _p(a, _u8p))
        return a

    def set_crc_matrix(self, m):
        m = np.ascontiguousarray(m, np.uint8)
        assert m.shape == (self.crc, self.K)
        if self.crc:
            self._call("set_crc_matrix", _p(m, _u8p))

    def set_tables(self, frozen, order):
        frozen = np.ascontiguousarray(frozen, np.uint8)
        order = np.ascontiguousarray(order, np.uint16)
        self._call("set_tables", _p(frozen, _u8p), _p(order, _u16p))

    def encode(self, info):
        info = np.ascontiguousarray(info, np.uint8)
        out = np.zeros(self.N, np.uint8)
        self._call("encode", _p(info, _u8p), _p(out, _u8p))
        return out

    def decode_scl_llr(self, llr, L):
        llr = np.ascontiguousarray(llr, np.float64)
        single = llr.ndim == 1
        llr2 = llr.reshape(-1, self.N)
        B = llr2.shape[0]
        out = np.zeros((B, self.K), np.uint8)
        self._call("decode_scl_llr_batch", _p(llr2, _dp), C.c_long(B), C.c_int(L), _p(out, _u8p))
        return out[0] if single else out

    def decode_scl_p1(self, p1, p0, L):
        p1 = np.ascontiguousarray(p1, np.float64)
        p0 = np.ascontiguousarray(p0, np.float64)
        out = np.zeros(self.K, np.uint8)
        self._call("decode_scl_p1", _p(p1, _dp), _p(p0, _dp), C.c_int(L), _p(out, _u8p))
        return out


class Reference(_Base):
    """The unmodified PolarC (oracle/_ref/libpolarc_ref.so). Present only where it was built."""
    prefix = "ref_"

    def __init__(self, n, K, eps, crc, srand=None):
        lib = _load(REF_SO)
        if lib is None:
            raise FileNotFoundError(REF_SO)
        if srand is not None:
            lib.ref_srand(C.c_uint(srand))
        super().__init__(lib, n, K, eps, crc)

    def get_bler_quick(self, ebno, Ls):
        ebno = np.ascontiguousarray(ebno, np.float64)
        Ls = np.ascontiguousarray(Ls, np.uint8)
        out = np.zeros((len(Ls), len(ebno)), np.float64)
        self._call("get_bler_quick", _p(ebno, _dp), C.c_int(len(ebno)), _p(Ls, _u8p), C.c_int(len(Ls)), _p(out, _dp))
        return out


def have_reference():
    return os.path.exists(REF_SO)


class Oracle(_Base):
    """Our CPU restatement (oracle/liboracle.so)."""
    prefix = "orc_"

    def __init__(self, n, K, eps, crc, srand=None):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        if srand is not None:
            C.CDLL(None).srand(C.c_uint(srand))
        super().__init__(lib, n, K, eps, crc)

    def decode_scl_llr_pm(self, llr, L):
        llr = np.ascontiguousarray(llr, np.float64)
        out = np.zeros(self.K, np.uint8)
        pm = C.c_double(0)
        self._call("decode_scl_llr", _p(llr, _dp), C.c_int(L), _p(out, _u8p), C.byref(pm), restype=C.c_int)
        return out, pm.value

    def decode_sc_p1(self, p1):
        p1 = np.ascontiguousarray(p1, np.float64)
        out = np.zeros(self.K, np.float64)
        self._call("decode_sc_p1", _p(p1, _dp), _p(out, _dp))
        return out

    def get_bler_quick_ref(self, ebno, Ls, max_runs=1000, max_err=100):
        ebno = np.ascontiguousarray(ebno, np.float64)
        Ls = np.ascontiguousarray(Ls, np.uint8)
        out = np.zeros((len(Ls), len(ebno)), np.float64)
        self._call("get_bler_quick_ref", _p(ebno, _dp), C.c_int(len(ebno)), _p(Ls, _u8p), C.c_int(len(Ls)),
                   C.c_int(max_runs), C.c_int(max_err), _p(out, _dp))
        return out

    def snr_sqrt_linear(self, ebno_db):
        return self._call("snr_sqrt_linear", C.c_double(ebno_db), restype=C.c_double)

    def synth_llr(self, seed, trial0, B, s):
        llr = np.zeros((B, self.N), np.float64)
        info = np.zeros((B, self.K), np.uint8)
        self._call("synth_llr_batch", C.c_uint64(seed), C.c_uint64(trial0), C.c_long(B), C.c_double(s),
                   _p(llr, _dp), _p(info, _u8p))
        return llr, info

    def synth_bicm_llr(self, cid, seed, trial0, B, snr_db):
        llr = np.zeros((B, self.N), np.float64)
        info = np.zeros((B, self.K), np.uint8)
        self._call("synth_bicm_llr_batch", C.c_int(cid), C.c_uint64(seed), C.c_uint64(trial0), C.c_long(B),
                   C.c_double(snr_db), _p(llr, _dp), _p(info, _u8p))
        return llr, info

    def mc_batch(self, seed, t0, T, stride, ebno, Ls, enabled, err, run):
        ebno = np.ascontiguousarray(ebno, np.float64)
        Ls = np.ascontiguousarray(Ls, np.uint8)
        enabled = np.ascontiguousarray(enabled, np.uint8)
        assert err.dtype == np.uint64 and run.dtype == np.uint64
        self._call("mc_batch", C.c_uint64(seed), C.c_uint64(t0), C.c_long(T), C.c_long(stride), _p(ebno, _dp), C.c_int(len(ebno)),
                   _p(Ls, _u8p), C.c_int(len(Ls)), _p(enabled, _u8p), _p(err, _u64p), _p(run, _u64p))


    def mc_batch_bicm(self, cid, seed, t0, T, stride, snr_db, Ls, enabled, err, run):
        snr = np.ascontiguousarray(snr_db, np.float64)
        Ls = np.ascontiguousarray(Ls, np.uint8)
        enabled = np.ascontiguousarray(enabled, np.uint8)
        self._call("mc_batch_bicm", C.c_int(cid), C.c_uint64(seed), C.c_uint64(t0), C.c_long(T), C.c_long(stride),
                   _p(snr, _dp), C.c_int(len(snr)), _p(Ls, _u8p), C.c_int(len(Ls)), _p(enabled, _u8p),
                   _p(err, _u64p), _p(run, _u64p))


def mc_construction(n, cid, design_snr_db, seed, trial0, num_runs):
    """oracle/polar_oracle.c orc_mc_construction: per-position genie-SC error counts."""
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    lib = C.CDLL(ORACLE_SO)
    cnt = np.zeros(1 << n, np.uint64)
    lib.orc_mc_construction(C.c_int(n), C.c_int(cid), C.c_double(design_snr_db), C.c_uint64(seed), C.c_uint64(trial0),
                            C.c_long(num_runs), _p(cnt, _u64p))
    return cnt


def mc_construction_run(n, cid, design_snr_db, seed, trial):
    """One run: (p1[N], error flags[N])."""
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    lib = C.CDLL(ORACLE_SO)
    N = 1 << n
    p1 = np.zeros(N, np.float64)
    cnt = np.zeros(N, np.uint64)
    lib.orc_mc_construction_run(C.c_int(n), C.c_int(cid), C.c_double(design_snr_db), C.c_uint64(seed), C.c_uint64(trial),
                                _p(p1, _dp), _p(cnt, _u64p))
    return p1, cnt.astype(np.uint8)
