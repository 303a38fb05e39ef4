"""A stand-in for the MATLAB side of the MEX boundary (MATLAB is not in the image): builds
polar_amd/matlab/polar_mex.cpp against the WORKING mx / mex runtime of tests/mex_runtime/ into a shared library, and drives
its ``mexFunction`` the way MATLAB would — arguments as column-major mxArrays of MATLAB's classes, outputs read back as
column-major arrays, ``mexErrMsgIdAndTxt`` surfacing as an exception with its identifier.

``PolarCodeM`` is polar_amd/matlab/PolarCode.m re-expressed call for call (same gateway commands, same argument classes and
order, same post-processing) so that PolarM/main.m:4-12 can be run as a test. Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.path.join(ROOT, "tests", "mex_runtime")
SO = os.path.join(RT, "_build", "libpolar_mex_fake.so")

# matrix.h class ids (tests/mex_runtime/mex.h)
CHAR, DOUBLE, SINGLE, INT8, UINT8, INT16, UINT16, INT32, UINT32, INT64, UINT64 = 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
_CLS = {np.dtype(np.float64): DOUBLE, np.dtype(np.float32): SINGLE, np.dtype(np.int8): INT8, np.dtype(np.uint8): UINT8,
        np.dtype(np.int16): INT16, np.dtype(np.uint16): UINT16, np.dtype(np.int32): INT32, np.dtype(np.uint32): UINT32,
        np.dtype(np.int64): INT64, np.dtype(np.uint64): UINT64}
_DT = {v: k for k, v in _CLS.items()}
_DT[CHAR] = np.dtype(np.uint16)


class MexError(RuntimeError):
    """mexErrMsgIdAndTxt(id, msg)."""

    def __init__(self, ident, msg):
        super().__init__(f"{ident}: {msg}")
        self.identifier, self.message = ident, msg


def build(force=False):
    """g++: the gateway + the fake runtime, linked against the in-tree libpolar_amd.so (what `mex ... -lpolar_amd` does)."""
    from polar_amd import build as pb
    lib = pb.build()
    srcs = [os.path.join(ROOT, "polar_amd", "matlab", "polar_mex.cpp"), os.path.join(RT, "mex_runtime.cpp")]
    deps = srcs + [os.path.join(RT, "mex.h"), os.path.join(ROOT, "polar_amd", "matlab", "polar_mex_layout.h"),
                   os.path.join(ROOT, "include", "polar_amd.h"), lib]
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        libdir = os.path.dirname(lib)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-I", RT, "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "polar_amd", "matlab")] + srcs +
                              ["-L", libdir, "-lpolar_amd", "-Wl,-rpath,$ORIGIN/../../../polar_amd", "-o", SO])
    return SO


class MxValue:
    """An argument converted once (Mex.value) and passed to several calls: a MATLAB variable that lives in the workspace while a
    script calls the MEX file repeatedly. Freed by .free()."""

    def __init__(self, mex, ptr):
        self.mex, self.ptr = mex, ptr

    def free(self):
        if self.ptr:
            self.mex.L.fm_free(self.ptr)
            self.ptr = None


class Mex:
    """``polar_mex = Mex(); out = polar_mex('cmd', args..., nlhs=k)``: one MEX file loaded into this process."""

    def __init__(self):
        import polar_amd
        polar_amd.lib()                     # (torch's HIP runtime and libpolar_amd.so first: one HIP runtime per process)
        L = C.CDLL(build())
        L.fm_array.restype = C.c_void_p
        L.fm_array.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
        L.fm_string.restype = C.c_void_p
        L.fm_string.argtypes = [C.c_char_p]
        L.fm_class.argtypes = [C.c_void_p]
        L.fm_m.restype = C.c_size_t
        L.fm_m.argtypes = [C.c_void_p]
        L.fm_n.restype = C.c_size_t
        L.fm_n.argtypes = [C.c_void_p]
        L.fm_data.restype = C.c_void_p
        L.fm_data.argtypes = [C.c_void_p]
        L.fm_free.argtypes = [C.c_void_p]
        L.fm_live.restype = C.c_long
        L.fm_call.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_size_t]
        self.L = L

    # ---- MATLAB value -> mxArray --------------------------------------------------------------------------------------
    def value(self, v):
        return MxValue(self, self._to_mx(v))

    def _to_mx(self, v):
        if isinstance(v, MxValue):
            return v.ptr
        if isinstance(v, str):
            return self.L.fm_string(v.encode())
        if isinstance(v, (bool, int, float, np.integer, np.floating)) and not isinstance(v, np.generic):
            v = np.float64(v)                                  # a MATLAB literal is a double scalar
        a = np.asarray(v)
        if a.dtype not in _CLS:
            raise TypeError(f"no MATLAB class for dtype {a.dtype}")
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(1, -1)                               # MATLAB vectors written x(:)' are 1 x N
        elif a.ndim != 2:
            raise TypeError("only 2-D arrays")
        f = np.asfortranarray(a)                               # column-major storage
        p = self.L.fm_array(_CLS[a.dtype], a.shape[0], a.shape[1], f.ctypes.data_as(C.c_void_p) if f.size else None)
        if not p:
            raise MemoryError("fm_array")
        return p

    def _from_mx(self, p):
        cls, m, n = self.L.fm_class(p), self.L.fm_m(p), self.L.fm_n(p)
        dt = _DT[cls]
        if m * n == 0:
            return np.zeros((m, n), dt)
        buf = (C.c_char * (m * n * dt.itemsize)).from_address(self.L.fm_data(p))
        return np.frombuffer(buf, dt).reshape((m, n), order="F").copy(order="F")      # (MATLAB arrays stay column-major)

    def __call__(self, cmd, *args, nlhs=1):
        import time
        prhs = [self._to_mx(cmd)] + [self._to_mx(a) for a in args]
        owned = [p for p, a in zip(prhs, (cmd,) + args) if not isinstance(a, MxValue)]
        arr = (C.c_void_p * len(prhs))(*prhs)
        out = (C.c_void_p * max(nlhs, 1))()
        eid, emsg = C.create_string_buffer(256), C.create_string_buffer(1024)
        try:
            t0 = time.perf_counter()
            rc = self.L.fm_call(nlhs, out, len(prhs), arr, eid, emsg, 1024)
            self.last_call_seconds = time.perf_counter() - t0      # inside mexFunction: what the MATLAB caller waits for
            if rc:
                raise MexError(eid.value.decode(), emsg.value.decode())
            res = []
            for i in range(nlhs):
                if not out[i]:
                    raise MexError("fake_mx:output", f"output {i + 1} of {nlhs} was not assigned by '{cmd}'")
                res.append(self._from_mx(out[i]))
            return res[0] if nlhs == 1 else tuple(res) if nlhs else None
        finally:
            for p in owned:
                self.L.fm_free(p)
            for i in range(nlhs):
                if out[i]:
                    self.L.fm_free(out[i])

    def live_arrays(self):
        return self.L.fm_live()

    def lock_depth(self):
        return self.L.fm_lock_depth()


_mex = None


def polar_mex():
    global _mex
    if _mex is None:
        _mex = Mex()
    return _mex


class PolarCodeM:
    """polar_amd/matlab/PolarCode.m, method for method (line numbers of that file in the comments). Row vectors of doubles in
    and out like the reference class; info_bits 1-based."""

    def __init__(self, block_length, info_length, design_epsilon, crc_size=0):          # PolarCode.m:26-43
        self.mex = polar_mex()
        self.block_length, self.info_length, self.crc_size = block_length, info_length, crc_size
        self.n = float(np.log2(block_length))
        self.design_epsilon = design_epsilon
        self.h = self.mex('create', self.n, info_length, design_epsilon, crc_size)
        fz, order, crcm = self.mex('tables', self.h, nlhs=3)
        self.frozen_bits = fz.reshape(1, -1).astype(np.float64)
        self.info_bits = order.reshape(-1)[:info_length + crc_size].astype(np.float64) + 1
        self.crc_matrix = crcm.astype(np.float64)
        self.cc_method, self.cc_parameter, self.cc_misc = 'bhattacharya', design_epsilon, ''

    @staticmethod
    def _num2str(v):                       # MATLAB num2str of the values the unique string sees (integers, short decimals)
        return str(int(v)) if float(v) == int(v) else f"{v:.4g}"

    def get_unique_string(self):           # :67-70
        return f"{self._num2str(self.block_length)}_{len(self.info_bits)}_cc_method_{self.cc_method}_cc_param_{self._num2str(self.cc_parameter)}_{self.cc_misc}"

    def monte_carlo_code_construction(self, design_snr_db, num_runs=100e3, constellation_name='bpsk', receiver_algo='bicm', seed=1, data_dir='CodeConstructionData'):   # :44-66
        assert receiver_algo == 'bicm'
        cid = ['ask4-gray', 'ask8-gray', 'ask16-gray', 'bpsk'].index(constellation_name) + 1
        self.cc_method, self.cc_parameter = 'monte-carlo', design_snr_db
        self.cc_misc = f"{constellation_name}_{receiver_algo}_{self._num2str(num_runs)}"
        table_file = os.path.join(data_dir, 'MC_block_length_' + self.get_unique_string() + '.txt')
        old = self.h
        self.h, fz, order0, est = self.mex('monte_carlo_design', self.n, self.info_length, self.crc_size, self.crc_matrix.astype(np.uint8), cid,
                                           design_snr_db, num_runs, seed, table_file, nlhs=4)
        self.mex('destroy', old, nlhs=0)
        self.frozen_bits = fz.astype(np.float64)
        self.info_bits = order0.reshape(-1)[:self.info_length + self.crc_size].astype(np.float64) + 1
        return float(est[0, 0]), table_file

    def delete(self):                      # :71-76
        if self.h is not None:
            self.mex('destroy', self.h, nlhs=0)
            self.h = None

    def encode(self, info_bits):           # :77-79
        return self.mex('encode', self.h, np.asarray(info_bits).reshape(-1).astype(np.uint8)).astype(np.float64)

    def decode_sc_p1(self, p1):            # :80-82
        return self.mex('decode_sc_p1', self.h, np.asarray(p1, np.float64).reshape(-1)).astype(np.float64)

    def decode_scl_p1(self, p1, p0, list_size):      # :83-85
        return self.mex('decode_scl_p1', self.h, np.asarray(p1, np.float64).reshape(-1), np.asarray(p0, np.float64).reshape(-1), list_size).astype(np.float64)

    def decode_scl_llr(self, llr, list_size, layout=None):      # :86-98
        llr = np.asarray(llr)
        if llr.dtype != np.float32:
            llr = llr.astype(np.float64)
        if layout is None:
            return self.mex('decode_scl_llr', self.h, llr, list_size).astype(np.float64)
        return self.mex('decode_scl_llr', self.h, llr, list_size, layout).astype(np.float64)

    def get_bler_quick(self, ebno_vec, list_size_vec, max_runs=500, max_err=50, seed=1, devices=(), constellation_id=0):     # :99-113
        b, e = self.mex('get_bler_quick', self.h, np.asarray(ebno_vec, np.float64).reshape(-1), np.asarray(list_size_vec).reshape(-1).astype(np.uint8),
                        max_runs, max_err, seed, np.asarray(devices, np.int32).reshape(1, -1), constellation_id, nlhs=2)
        return b.T, e.T
