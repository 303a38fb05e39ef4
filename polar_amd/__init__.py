"""polar_amd — MI355X-native polar SC/SCL decoder: Python host mirror of the reference's
``PolarCode`` class surface (PolarC/PolarCode.h:19-34, PolarM/PolarCode.m:59-93,266-322,781-850)
over the C-ABI of include/polar_amd.h.

This module is plumbing only: every compute call goes through ``libpolar_amd.so`` (hand-written
HIP kernels for gfx950). There is NO CPU fallback: if the shared library is missing, or no HIP
device is usable, the calls raise.
"""
import ctypes as C
import os

import numpy as np

try:  # must precede loading libpolar_amd.so so both share ONE HIP runtime (see build.py)
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing; the library also works stand-alone
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POLAR_AMD_LIB") or os.path.join(_HERE, "libpolar_amd.so")   # (override: A/B builds)

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u64p = C.POINTER(C.c_uint64)


ASK4_GRAY, ASK8_GRAY, ASK16_GRAY, BPSK = 1, 2, 3, 4   # include/polar_synth.h POLAR_CONST_*
CONSTELLATION_NAMES = {"bpsk": BPSK, "ask4-gray": ASK4_GRAY, "ask8-gray": ASK8_GRAY, "ask16-gray": ASK16_GRAY}   # Constellation.m:41-59


def _constellation_id(c):
    if isinstance(c, str):
        if c not in CONSTELLATION_NAMES:
            raise PolarError(f"unsupported constellation {c!r} (supported: {sorted(CONSTELLATION_NAMES)})")
        return CONSTELLATION_NAMES[c]
    return int(c)


class PolarError(RuntimeError):
    pass


class PolarWeakLeavesWarning(UserWarning):
    """polar_create_explicit returned POLAR_W_WEAK_LEAVES (include/polar_amd.h)."""


POLAR_W_WEAK_LEAVES = 1


_lib = None


def lib():
    """Load libpolar_amd.so (built by polar_amd.build.build()); raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PolarError(
                f"{LIB_PATH} not found: build it with `python -m polar_amd.build` "
                "(the HIP extension is mandatory; there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.polar_last_error.restype = C.c_char_p
        L.polar_snr_sqrt_linear.restype = C.c_double
        L.polar_snr_sqrt_linear.argtypes = [C.c_void_p, C.c_double]
        _lib = L
    return _lib


def use_library(path=None):
    """Load another build of the library for everything created from now on (None = the product again). Handles belong to the
    library that made them: the caller keeps none across the switch. The tests of the failure protocol load the test build
    (libpolar_amd_test.so: the fault-injection hooks of include/polar_amd_debug.h exist only there)."""
    global _lib, LIB_PATH
    LIB_PATH = path or os.environ.get("POLAR_AMD_LIB") or os.path.join(_HERE, "libpolar_amd.so")
    _lib = None
    return lib()


def _check(rc):
    """Negative = error; positive = a non-error status (POLAR_W_WEAK_LEAVES), returned to the caller."""
    if rc < 0:
        raise PolarError(f"polar_amd error {rc}: {lib().polar_last_error().decode()}")
    return rc


def _p(a, t):
    return a.ctypes.data_as(t)


def _stream_ptr(stream):
    if stream is None:
        if torch is not None and torch.cuda.is_available():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return C.c_void_p(0)
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


class PolarCode:
    """Drop-in for the reference ``PolarCode``.

    ``PolarCode(num_layers, info_length, epsilon, crc_size)`` follows PolarC (PolarCode.h:19);
    ``PolarCode.from_block_length(block_length, info_length, design_epsilon, crc_size=0)`` follows
    PolarM's argument order (PolarCode.m:59). ``PolarCode.from_tables`` takes explicit tables.
    """

    def __init__(self, num_layers, info_length, epsilon, crc_size=0, _handle=None):
        L = self._L = lib()          # (a handle belongs to the library that made it: use_library() may switch the default later)
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            self._chk(L.polar_create(C.c_int(num_layers), C.c_int(info_length), C.c_double(epsilon),
                                  C.c_int(crc_size), C.byref(self._h)))
        n, N, K, crc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._chk(L.polar_get_params(self._h, C.byref(n), C.byref(N), C.byref(K), C.byref(crc)))
        self.n, self.block_length, self.info_length, self.crc_size = n.value, N.value, K.value, crc.value
        self.N, self.K = self.block_length, self.info_length

    @property
    def weak_leaves(self):
        """Unfrozen leaves the handle classified as weak at creation (polar_get_weak_leaves)."""
        return int(self._L.polar_get_weak_leaves(self._h))

    def debug_set(self, key, value):
        """Measurement knobs / test hooks of include/polar_amd_debug.h (polar_debug_set)."""
        self._chk(self._L.polar_debug_set(self._h, key.encode(), C.c_long(int(value))))

    def debug_get(self, key):
        f = self._L.polar_debug_get
        f.restype = C.c_long
        return int(f(self._h, key.encode()))

    @classmethod
    def from_block_length(cls, block_length, info_length, design_epsilon, crc_size=0):
        n = int(round(np.log2(block_length)))
        if (1 << n) != block_length:
            raise PolarError("block_length must be a power of two")
        return cls(n, info_length, design_epsilon, crc_size)

    @classmethod
    def from_tables(cls, num_layers, info_length, crc_size, frozen, order, crc_matrix=None):
        frozen = np.ascontiguousarray(frozen, np.uint8)
        order = np.ascontiguousarray(order, np.uint16)
        N = 1 << num_layers
        if frozen.shape != (N,) or order.shape != (N,):
            raise PolarError("frozen/order must have N entries")
        cm = None
        if crc_size:
            cm = np.ascontiguousarray(crc_matrix, np.uint8)
            if cm.shape != (crc_size, info_length):
                raise PolarError("crc_matrix must be crc x K")
        h = C.c_void_p()
        status = _check(lib().polar_create_explicit(C.c_int(num_layers), C.c_int(info_length), C.c_int(crc_size),
                                                    _p(frozen, _u8p), _p(order, _u16p),
                                                    _p(cm, _u8p) if cm is not None else None, C.byref(h)))
        code = cls(num_layers, info_length, float("nan"), crc_size, _handle=h)
        if status == POLAR_W_WEAK_LEAVES:      # a valid handle; bit-exactness with the reference is limited (polar_amd.h)
            import warnings
            warnings.warn(f"polar_amd: {lib().polar_last_error().decode()}", PolarWeakLeavesWarning, stacklevel=2)
        return code

    @classmethod
    def from_construction_file(cls, path, info_length, crc_size=0, crc_matrix=None):
        """Code from a PolarM Monte-Carlo construction file (PolarM/CodeConstructionData/*.txt: one
        per-channel error count per line, written at PolarCode.m:120-124). As PolarCode.m:126-135:
        stable ascending sort of the counts, the first K+crc positions are the unfrozen set and their
        order is the info-bit order."""
        return cls.from_counts(np.loadtxt(path).reshape(-1), info_length, crc_size, crc_matrix)

    @classmethod
    def from_counts(cls, counts, info_length, crc_size=0, crc_matrix=None):
        """Code from a per-channel error-count table (PolarCode.m:126-135): stable ascending sort,
        the first K+crc positions are the unfrozen set and their order is the info-bit order."""
        counts = np.asarray(counts).reshape(-1)
        N = counts.size
        n = int(round(np.log2(N)))
        if (1 << n) != N:
            raise PolarError("the count table must hold a power-of-two number of entries")
        order = np.argsort(counts, kind="stable").astype(np.uint16)
        frozen = np.ones(N, np.uint8)
        frozen[order[: info_length + crc_size]] = 0
        return cls.from_tables(n, info_length, crc_size, frozen, order, crc_matrix)

    @classmethod
    def from_monte_carlo(cls, block_length, info_length, design_snr_db, crc_size=0, num_runs=100000,
                         constellation_name="bpsk", receiver_algo="bicm", seed=1, crc_matrix=None, data_dir=None):
        """Code designed by PolarM's `monte_carlo_code_construction` (PolarCode.m:95-141): same argument
        meaning and defaults; the genie-aided SC runs on the GPU (mc_construction).
        With ``data_dir`` the table is read from / written to
        ``MC_block_length_<unique string>.txt`` exactly as the reference does (:111-124)."""
        if receiver_algo != "bicm":
            raise PolarError("only the 'bicm' receiver is built (SURVEY §2: the MLC demapper is out of scope)")
        path = None
        if data_dir is not None:
            path = os.path.join(data_dir, "MC_block_length_" + construction_unique_string(
                block_length, info_length + crc_size, design_snr_db, constellation_name, receiver_algo, num_runs) + ".txt")
        if path is not None and os.path.exists(path):
            counts = np.loadtxt(path).reshape(-1)
        else:
            n = int(round(np.log2(block_length)))
            if (1 << n) != block_length:
                raise PolarError("block_length must be a power of two")
            counts = mc_construction(n, design_snr_db, num_runs, constellation_name, seed=seed)
            if path is not None:
                write_construction_file(path, counts)
        if crc_size and crc_matrix is None:      # PolarCode.m:83: crc_matrix = floor(2*rand(crc_size, info_length))
            crc_matrix = np.random.default_rng(seed).integers(0, 2, (crc_size, info_length)).astype(np.uint8)
        code = cls.from_counts(counts, info_length, crc_size, crc_matrix)
        code.construction_counts = np.asarray(counts)
        # PolarCode.m:136: bler_estimate = sum(channels(info_bits)) / num_runs
        code.bler_estimate = float(np.sort(np.asarray(counts, np.float64), kind="stable")[: info_length + crc_size].sum() / num_runs)
        return code

    def monte_carlo_code_construction(self, design_snr_db, num_runs=100000, constellation_name="bpsk",
                                      receiver_algo="bicm", seed=1, data_dir=None):
        """In-place redesign of this code, as the reference method of the same name
        (PolarCode.m:95-141): block length, K, crc size and the crc matrix are kept, the frozen set and
        info-bit order are replaced."""
        cm = self.crc_matrix if self.crc_size else None
        new = PolarCode.from_monte_carlo(self.block_length, self.info_length, design_snr_db, self.crc_size, num_runs,
                                         constellation_name, receiver_algo, seed, cm, data_dir)
        self.close()
        self._h, new._h = new._h, None
        self.construction_counts, self.bler_estimate = new.construction_counts, new.bler_estimate
        return self.bler_estimate

    def _chk(self, rc):
        if rc < 0:
            raise PolarError(f"polar_amd error {rc}: {self._L.polar_last_error().decode()}")
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._L.polar_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables ------------------------------------------------------------------------
    @property
    def frozen_bits(self):
        a = np.zeros(self.N, np.uint8)
        self._chk(self._L.polar_get_frozen(self._h, _p(a, _u8p)))
        return a

    @property
    def channel_order_descending(self):
        a = np.zeros(self.N, np.uint16)
        self._chk(self._L.polar_get_order(self._h, _p(a, _u16p)))
        return a

    @property
    def bit_rev_order(self):
        a = np.zeros(self.N, np.uint16)
        self._chk(self._L.polar_get_bitrev(self._h, _p(a, _u16p)))
        return a

    @property
    def crc_matrix(self):
        a = np.zeros((self.crc_size, self.K), np.uint8)
        if self.crc_size:
            self._chk(self._L.polar_get_crc_matrix(self._h, _p(a, _u8p)))
        return a

    @crc_matrix.setter
    def crc_matrix(self, m):
        m = np.ascontiguousarray(m, np.uint8)
        if m.shape != (self.crc_size, self.K):
            raise PolarError("crc_matrix must be crc x K")
        if self.crc_size:
            self._chk(self._L.polar_set_crc_matrix(self._h, _p(m, _u8p)))

    def reserve(self, B, list_size):
        """Pre-size the device scratch for decodes of up to B codewords at this list size (polar_reserve)."""
        self._chk(self._L.polar_reserve(self._h, C.c_long(B), C.c_int(list_size)))

    def set_tuning(self, waves_per_cu=0, lds_log=0):
        self._chk(self._L.polar_set_tuning(self._h, C.c_int(waves_per_cu), C.c_int(lds_log)))

    def set_mode(self, mode=0):
        """Node arithmetic of decode_scl_llr: 0 automatic, 1 LLR-domain kernel, 2 exp-domain kernel (+ fallback pass)."""
        self._chk(self._L.polar_set_mode(self._h, C.c_int(mode)))

    def snr_sqrt_linear(self, ebno_db):
        return self._L.polar_snr_sqrt_linear(self._h, C.c_double(ebno_db))

    # ---- encode (PolarCode.cpp:60-91) ---------------------------------------------------
    def encode(self, info_bits):
        info = np.ascontiguousarray(info_bits, np.uint8)
        single = info.ndim == 1
        info2 = info.reshape(-1, self.K)
        out = np.zeros((info2.shape[0], self.N), np.uint8)
        self._chk(self._L.polar_encode_batch(self._h, _p(info2, _u8p), C.c_long(info2.shape[0]), _p(out, _u8p)))
        return out[0] if single else out

    # ---- decoders -----------------------------------------------------------------------
    def decode_scl_llr(self, llr, list_size, out=None):
        """PolarCode::decode_scl_llr (PolarCode.cpp:130-148). `llr` is [N] or [B, N]; float32 arrays
        travel as float32 and are widened exactly on the device, anything else is taken as float64.
        out (optional): a C-contiguous uint8 [B, K] array to receive the bits (a caller that decodes batch after batch keeps
        one: a fresh 64-MiB array costs its page faults on every call)."""
        f32 = isinstance(llr, np.ndarray) and llr.dtype == np.float32
        a = np.ascontiguousarray(llr) if f32 else np.ascontiguousarray(llr, np.float64)
        single = a.ndim == 1
        a2 = a.reshape(-1, self.N)
        if out is None:
            out = np.zeros((a2.shape[0], self.K), np.uint8)
        elif out.dtype != np.uint8 or out.shape != (a2.shape[0], self.K) or not out.flags.c_contiguous:
            raise PolarError("out must be a C-contiguous uint8 array of shape [B, K]")
        if f32:
            self._chk(self._L.polar_decode_scl_llr_batch_f32(self._h, _p(a2, C.POINTER(C.c_float)), C.c_long(a2.shape[0]),
                                                        C.c_int(list_size), _p(out, _u8p)))
        else:
            self._chk(self._L.polar_decode_scl_llr_batch(self._h, _p(a2, _dp), C.c_long(a2.shape[0]), C.c_int(list_size),
                                                    _p(out, _u8p)))
        return out[0] if single else out

    def decode_scl_llr_dev_f32(self, llr_ptr, B, list_size, out_ptr, pm_ptr=0, stream=None):
        """Device-resident float32 LLRs [B, N] -> uint8 [B, K]; asynchronous on `stream`."""
        self._chk(self._L.polar_decode_scl_llr_batch_dev_f32(self._h, C.c_void_p(llr_ptr), C.c_long(B), C.c_int(list_size),
                                                        C.c_void_p(out_ptr), C.c_void_p(pm_ptr), _stream_ptr(stream)))

    def decode_scl_p1(self, p1, p0, list_size):
        """PolarCode::decode_scl_p1 (PolarCode.cpp:110-128)."""
        a = np.ascontiguousarray(p1, np.float64)
        b = np.ascontiguousarray(p0, np.float64)
        single = a.ndim == 1
        a2, b2 = a.reshape(-1, self.N), b.reshape(-1, self.N)
        out = np.zeros((a2.shape[0], self.K), np.uint8)
        self._chk(self._L.polar_decode_scl_p1_batch(self._h, _p(a2, _dp), _p(b2, _dp), C.c_long(a2.shape[0]),
                                               C.c_int(list_size), _p(out, _u8p)))
        return out[0] if single else out

    def decode_sc_p1(self, p1):
        """PolarM decode_sc_p1 (PolarCode.m:290-295)."""
        a = np.ascontiguousarray(p1, np.float64)
        single = a.ndim == 1
        a2 = a.reshape(-1, self.N)
        out = np.zeros((a2.shape[0], self.K), np.float64)
        self._chk(self._L.polar_decode_sc_p1_batch(self._h, _p(a2, _dp), C.c_long(a2.shape[0]), _p(out, _dp)))
        return out[0] if single else out

    # names used by BASELINE.json's north_star
    decode_SCL_LLR = decode_scl_llr
    decode_SCL_P1 = decode_scl_p1
    decode_SC_P1 = decode_sc_p1

    # ---- device-resident entry points (torch tensors are only memory + stream plumbing) ---
    def decode_scl_llr_dev(self, llr_ptr, B, list_size, out_ptr, pm_ptr=0, stream=None, ev_start=0, ev_stop=0):
        """ev_start / ev_stop: raw hipEvent_t handles (e.g. torch.cuda.Event(...).cuda_event) recorded
        immediately around the dominant kernel's launch."""
        self._chk(self._L.polar_decode_scl_llr_batch_dev_ev(self._h, C.c_void_p(llr_ptr), C.c_long(B), C.c_int(list_size),
                                                       C.c_void_p(out_ptr), C.c_void_p(pm_ptr), _stream_ptr(stream),
                                                       C.c_void_p(ev_start), C.c_void_p(ev_stop)))

    def synth_llr_dev(self, seed, trial0, B, s, llr_ptr, info_ptr=0, stream=None):
        self._chk(self._L.polar_synth_llr_dev(self._h, C.c_uint64(seed), C.c_uint64(trial0), C.c_long(B), C.c_double(s),
                                         C.c_void_p(llr_ptr), C.c_void_p(info_ptr), _stream_ptr(stream)))

    def count_errors_dev(self, a_ptr, b_ptr, B, counter_ptr, stream=None):
        self._chk(self._L.polar_count_errors_dev(self._h, C.c_void_p(a_ptr), C.c_void_p(b_ptr), C.c_long(B),
                                            C.c_void_p(counter_ptr), _stream_ptr(stream)))

    def encode_dev(self, info_ptr, B, coded_ptr, stream=None):
        self._chk(self._L.polar_encode_batch_dev(self._h, C.c_void_p(info_ptr), C.c_long(B), C.c_void_p(coded_ptr),
                                            _stream_ptr(stream)))

    # ---- Monte-Carlo (PolarCode.cpp:658-785) ---------------------------------------------
    def mc_batch(self, seed, t0, T, stride, ebno_vec, list_size_vec, enabled, err, run):
        ebno = np.ascontiguousarray(ebno_vec, np.float64)
        Ls = np.ascontiguousarray(list_size_vec, np.uint8)
        enabled = np.ascontiguousarray(enabled, np.uint8)
        assert err.dtype == np.uint64 and run.dtype == np.uint64
        self._chk(self._L.polar_mc_batch(self._h, C.c_uint64(seed), C.c_uint64(t0), C.c_long(T), C.c_long(stride),
                                    _p(ebno, _dp), C.c_int(len(ebno)), _p(Ls, _u8p), C.c_int(len(Ls)),
                                    _p(enabled, _u8p), _p(err, _u64p), _p(run, _u64p)))

    def synth_bicm_llr_dev(self, constellation, seed, trial0, B, snr_db, llr_ptr, info_ptr=0, stream=None):
        self._chk(self._L.polar_synth_bicm_llr_dev(self._h, C.c_int(_constellation_id(constellation)), C.c_uint64(seed), C.c_uint64(trial0),
                                              C.c_long(B), C.c_double(snr_db), C.c_void_p(llr_ptr),
                                              C.c_void_p(info_ptr), _stream_ptr(stream)))

    def mc_batch_bicm(self, constellation, seed, t0, T, stride, snr_db_vec, list_size_vec, enabled, err, run):
        snr = np.ascontiguousarray(snr_db_vec, np.float64)
        Ls = np.ascontiguousarray(list_size_vec, np.uint8)
        enabled = np.ascontiguousarray(enabled, np.uint8)
        assert err.dtype == np.uint64 and run.dtype == np.uint64
        self._chk(self._L.polar_mc_batch_bicm(self._h, C.c_int(_constellation_id(constellation)), C.c_uint64(seed), C.c_uint64(t0), C.c_long(T),
                                         C.c_long(stride), _p(snr, _dp), C.c_int(len(snr)), _p(Ls, _u8p),
                                         C.c_int(len(Ls)), _p(enabled, _u8p), _p(err, _u64p), _p(run, _u64p)))

    def mc_batch_ber(self, seed, t0, T, stride, ebno_vec, list_size_vec, enabled, err, bit_err, run):
        ebno = np.ascontiguousarray(ebno_vec, np.float64)
        Ls = np.ascontiguousarray(list_size_vec, np.uint8)
        enabled = np.ascontiguousarray(enabled, np.uint8)
        assert err.dtype == np.uint64 and run.dtype == np.uint64 and bit_err.dtype == np.uint64
        self._chk(self._L.polar_mc_batch_ber(self._h, C.c_uint64(seed), C.c_uint64(t0), C.c_long(T), C.c_long(stride),
                                        _p(ebno, _dp), C.c_int(len(ebno)), _p(Ls, _u8p), C.c_int(len(Ls)),
                                        _p(enabled, _u8p), _p(err, _u64p), _p(bit_err, _u64p), _p(run, _u64p)))

    def get_bler_quick(self, ebno_vec, list_size_vec, max_runs=1000, max_err=100, seed=1, batch=None,
                       return_ber=False, devices=None, constellation=None, return_counters=False):
        """PolarCode::get_bler_quick: returns bler[len(list_size_vec)][len(ebno_vec)] (PolarCode.cpp:658-785);
        with return_ber=True also PolarM's second output ber (PolarCode.m:781,848), same layout.
        batch=None: the library picks the rounds (see polar_amd.h). devices=[...]: shard the trials over these GPUs
        of the node from this one process (polar_get_bler_quick_multi_ex, RCCL all-reduce of the counters, one per round).
        constellation="ask16-gray" (...): the ASK Gray + BICM front end with `ebno_vec` read as the SNR axis in dB
        (PolarM/main_MC_CC_Comparison.m:44-119). return_counters=True: additionally a dict with the raw counters
        err / run (uint64, same layout) and the number of rounds."""
        ebno = np.ascontiguousarray(ebno_vec, np.float64)
        Ls = np.ascontiguousarray(list_size_vec, np.uint8)
        shape = (len(Ls), len(ebno))
        out = np.zeros(shape, np.float64)
        ber = np.zeros(shape, np.float64)
        err = np.zeros(shape, np.uint64)
        run = np.zeros(shape, np.uint64)
        if batch is None:
            batch = 0
        cid = 0 if constellation is None else _constellation_id(constellation)
        if devices is not None:
            devs = np.ascontiguousarray(devices, np.int32)
            dptr, nd = devs.ctypes.data_as(C.POINTER(C.c_int)), len(devs)
        else:
            dptr, nd = None, 1
        used, rounds = C.c_int(0), C.c_long(0)
        self._chk(self._L.polar_get_bler_quick_multi_ex(self._h, C.c_int(cid), dptr, C.c_int(nd),
                                                   _p(ebno, _dp), C.c_int(len(ebno)), _p(Ls, _u8p), C.c_int(len(Ls)),
                                                   C.c_long(max_runs), C.c_long(max_err), C.c_uint64(seed), C.c_long(batch),
                                                   _p(out, _dp), _p(ber, _dp), _p(err, _u64p), _p(run, _u64p),
                                                   C.byref(rounds), C.byref(used)))
        self.last_used_rccl = bool(used.value)
        res = (out, ber) if return_ber else (out,)
        if return_counters:
            res = res + ({"err": err, "run": run, "rounds": int(rounds.value)},)
        return res[0] if len(res) == 1 else res

    def get_bler_quick_rank(self, ebno_vec, list_size_vec, rank, world, reduce, max_runs=1000, max_err=100, seed=1, batch=None,
                            constellation=None):
        """polar_get_bler_quick_rank: this process is `rank` of `world` sharing the sweep; `reduce(a)` must SUM the uint64 numpy
        array `a` in place over the ranks (called collectively after every step). Returns (bler, ber, counters) like
        get_bler_quick(..., return_ber=True, return_counters=True)."""
        ebno = np.ascontiguousarray(ebno_vec, np.float64)
        Ls = np.ascontiguousarray(list_size_vec, np.uint8)
        shape = (len(Ls), len(ebno))
        out, ber = np.zeros(shape, np.float64), np.zeros(shape, np.float64)
        err, run = np.zeros(shape, np.uint64), np.zeros(shape, np.uint64)
        rounds = C.c_long(0)
        failure = []

        @C.CFUNCTYPE(C.c_int, C.c_void_p, _u64p, C.c_int)
        def cb(_user, ptr, n):
            try:
                a = np.ctypeslib.as_array(ptr, shape=(n,))
                reduce(a)
                return 0
            except Exception as ex:          # (no exception may cross the C frames)
                failure.append(ex)
                return 1
        cid = 0 if constellation is None else _constellation_id(constellation)
        rc = self._L.polar_get_bler_quick_rank(self._h, C.c_int(cid), C.c_int(rank), C.c_int(world), cb, None,
                                             _p(ebno, _dp), C.c_int(len(ebno)), _p(Ls, _u8p), C.c_int(len(Ls)),
                                             C.c_long(max_runs), C.c_long(max_err), C.c_uint64(seed), C.c_long(batch or 0),
                                             _p(out, _dp), _p(ber, _dp), _p(err, _u64p), _p(run, _u64p), C.byref(rounds))
        if failure:
            raise failure[0]
        _check(rc)
        return out, ber, {"err": err, "run": run, "rounds": int(rounds.value), "steps": self.debug_get("round_us_count")}


# ---- Monte-Carlo code construction (PolarM/PolarCode.m:95-196) ---------------------------------
def _num2str(x):
    """MATLAB num2str for the values the reference puts in file names (integers and short decimals)."""
    return str(int(x)) if float(x) == int(x) else ("%.4f" % float(x)).rstrip("0").rstrip(".")


def construction_unique_string(block_length, num_info_bits, design_snr_db, constellation_name="bpsk",
                               receiver_algo="bicm", num_runs=20000):
    """get_unique_string (PolarCode.m:258-261) for cc_method 'monte-carlo' (:107-109)."""
    return (f"{_num2str(block_length)}_{_num2str(num_info_bits)}_cc_method_monte-carlo_cc_param_"
            f"{_num2str(design_snr_db)}_{constellation_name}_{receiver_algo}_{_num2str(num_runs)}")


def write_construction_file(path, counts):
    """One count per line, '%d \n' (PolarCode.m:120-124)."""
    with open(path, "w") as f:
        for c in np.asarray(counts).reshape(-1):
            f.write("%d \n" % int(c))


def mc_construction(num_layers, design_snr_db, num_runs, constellation="bpsk", seed=1, trial0=0, batch=0, out=None):
    """Per-position error counts of the genie-aided SC decoder over ``num_runs`` Monte-Carlo runs
    (PolarCode.m:143-196 `monte_carlo`, 'bicm' receiver), computed on the GPU. Returns uint64[N];
    with ``out`` the counts are ADDED to it (shards of one trial range, see montecarlo.py)."""
    N = 1 << num_layers
    if out is None:
        out = np.zeros(N, np.uint64)
    if out.dtype != np.uint64 or out.shape != (N,) or not out.flags.c_contiguous:
        raise PolarError("out must be a contiguous uint64[N] array")
    _check(lib().polar_mc_construction(C.c_int(num_layers), C.c_int(_constellation_id(constellation)),
                                       C.c_double(design_snr_db), C.c_uint64(seed), C.c_uint64(trial0),
                                       C.c_long(num_runs), C.c_long(batch), _p(out, _u64p)))
    return out
