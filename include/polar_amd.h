/* polar_amd.h — C-ABI of the MI355X-native polar SC/SCL decoder (drop-in boundary).
 *
 * The reference (tavildar/Polar) has no FFI: its boundary is the public surface of
 * `class PolarCode` (PolarC/PolarCode.h:19-34; MATLAB twin PolarM/PolarCode.m:59-93,
 * 266-322, 781-850).  Every entry point below replaces one member of that surface
 * (cited per function) with plain pointers and sizes, so that a MEX gateway, a cgo/ctypes
 * stub or the C++ header-only mirror in polar_amd/cpp/PolarCode.hpp can bind it
 * (INTEGRATION.md shows each binding).
 *
 * Conventions
 *   - all functions return 0 on success, a negative POLAR_E_* code otherwise (one positive, non-error status exists:
 *     POLAR_W_WEAK_LEAVES from polar_create_explicit);
 *     polar_last_error() returns a thread-local message. No exceptions cross the ABI.
 *   - the caller owns every buffer; the library never retains a pointer past the call.
 *   - "host" entry points take host pointers (H2D/D2H included); "_dev" entry points take
 *     device pointers resident in HBM plus a hipStream_t passed as void*.
 *   - LLR sign convention as the reference: llr = ln(p0/p1), positive => bit 0
 *     (PolarCode.cpp:752). Bits are one uint8_t per bit (0/1), as the reference.
 *   - a handle is bound to the HIP device that was current at creation (or, when none was visible
 *     then, at its first compute call); every entry point runs there and restores the caller's current
 *     device. Calls on one handle must be serialised by the caller (the reference object is not
 *     re-entrant either: PolarCode.h:56-68), and the asynchronous "_dev" calls of one handle must all be
 *     issued on ONE stream (or be ordered by the caller): they share the handle's device scratch.
 *     A "_dev" call may (re)allocate that scratch when the batch or list size grows, which synchronises
 *     the device once.
 *   - the decoders run ONLY on the GPU: without a usable HIP device they fail with
 *     POLAR_E_DEVICE. There is no CPU fallback in this library.
 */
#ifndef POLAR_AMD_H
#define POLAR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POLAR_OK 0
#define POLAR_E_ARG (-1)      /* invalid argument (NULL, size, L out of range, ...) */
#define POLAR_E_DEVICE (-2)   /* no HIP device / HIP runtime error */
#define POLAR_E_NOMEM (-3)
#define POLAR_E_UNSUPPORTED (-4)
#define POLAR_W_WEAK_LEAVES 1 /* polar_create_explicit only: the handle is valid, but the table leaves unfrozen leaves in the
                                 worst synthetic channels (see polar_set_mode): bit-exactness with the reference is limited there */

#define POLAR_MAX_N_LOG2 15   /* reference: uint16_t _block_length (PolarCode.h:40) */
#define POLAR_MAX_LIST 64     /* reference loops forever for L > 127 (uint8_t, PolarCode.cpp:525) */
#define POLAR_MAX_CRC 32

typedef struct polar_code polar_code_t;

const char *polar_last_error(void);
/* library/ABI version: major*10000 + minor*100 + patch */
int polar_version(void);

/* ---- construction: PolarCode::PolarCode + create_bit_rev_order + initialize_frozen_bits
 *      (PolarCode.h:19-28, PolarCode.cpp:17-58, 647-656).
 * Bhattacharyya/BEC construction with design parameter `eps`; the info order is produced by
 * the same libstdc++ std::sort call as the reference, and — exactly like the reference — the
 * random-parity "CRC" matrix consumes crc*K draws of the process-global glibc rand() stream
 * (PolarCode.cpp:51-56).  Use polar_create_explicit() to pass every table yourself. */
int polar_create(int n, int K, double eps, int crc, polar_code_t **out);

/* Explicit tables (e.g. a Monte-Carlo constructed code, PolarM/PolarCode.m:111-135):
 * frozen[N] (1 = frozen), order[N] (= _channel_order_descending; only the first K+crc
 * entries are used), crc_matrix[crc*K] row-major (may be NULL when crc == 0). */
int polar_create_explicit(int n, int K, int crc, const uint8_t *frozen, const uint16_t *order,
                          const uint8_t *crc_matrix, polar_code_t **out);
void polar_destroy(polar_code_t *h);

/* table getters (PolarCode.h:45-48) */
int polar_get_params(const polar_code_t *h, int *n, int *N, int *K, int *crc);
int polar_get_frozen(const polar_code_t *h, uint8_t *frozen /*[N]*/);
int polar_get_order(const polar_code_t *h, uint16_t *order /*[N]*/);
int polar_get_bitrev(const polar_code_t *h, uint16_t *bitrev /*[N]*/);
int polar_get_crc_matrix(const polar_code_t *h, uint8_t *m /*[crc*K]*/);
int polar_set_crc_matrix(polar_code_t *h, const uint8_t *m /*[crc*K]*/);

/* ---- PolarCode::encode (PolarCode.cpp:60-91; PolarCode.m:266-276) ---- */
int polar_encode(polar_code_t *h, const uint8_t *info /*[K]*/, uint8_t *coded /*[N]*/);
int polar_encode_batch(polar_code_t *h, const uint8_t *info /*[B*K]*/, long B, uint8_t *coded /*[B*N]*/);
int polar_encode_batch_dev(polar_code_t *h, const uint8_t *d_info, long B, uint8_t *d_coded, void *stream);

/* ---- PolarCode::decode_scl_llr (PolarCode.cpp:130-148; PolarCode.m:312-322) ----
 * 1 <= L <= POLAR_MAX_LIST.  out[K] = decoded information bits in the reference's order
 * (Info[_channel_order_descending[beta]], PolarCode.cpp:172-174). */
int polar_decode_scl_llr(polar_code_t *h, const double *llr /*[N]*/, int L, uint8_t *out /*[K]*/);
/* batched, row-major, codeword-contiguous: llr[B*N] -> out[B*K].
 * Small batches — the reference's own loops decode one codeword per call (PolarCode.cpp:756,
 * PolarM/main_MC_CC_Comparison.m:96) — take latency kernels with ONE codeword per wave and the decoder state in LDS: list size 1
 * up to 2048 codewords (N <= 4096), list sizes 2 .. 8 up to one codeword per CU while the state fits 160 KiB of LDS (N = 2048: lists
 * up to 4; N = 1024: up to 8). Same bits as the batch kernels (tests/: every such test runs both). Batches of at most 64
 * codewords at list size 1 are staged in pinned, device-mapped host memory (no DMA copies).
 * Large batches (from 32 MiB of LLRs at L = 1, 1 GiB at L = 2, half a GiB or one full round of resident waves at L = 3 .. 8, half a GiB or two rounds for larger lists: below that one copy in, one launch and one copy out is faster) are PIPELINED inside the call: chunks are copied from the caller's (pageable) memory into
 * pinned slots by a few host threads, moved on a copy stream and decoded on two or three decode lanes with their own scratch
 * (the handle keeps slots, streams, lanes and threads: 0.3 - 1.3 GiB of pinned and device memory after the first such call):
 * min(device rate, PCIe rate) minus one decode launch, whatever the batch size; device memory use is bounded by the slots,
 * not by B. Same bits as one decode of the whole batch (tests/test_gpu_parity.py: chunk boundaries). */
int polar_decode_scl_llr_batch(polar_code_t *h, const double *llr, long B, int L, uint8_t *out);
/* device-resident: d_llr/d_out live in HBM; asynchronous on `stream` (hipStream_t).
 * d_pm (optional, may be NULL) receives the winning path metric per codeword. */
int polar_decode_scl_llr_batch_dev(polar_code_t *h, const double *d_llr, long B, int L, uint8_t *d_out,
                                   double *d_pm, void *stream);
/* single-precision LLRs at the boundary (half the PCIe / HBM input bytes): every float is widened
 * exactly to the reference's double on the device, so the result equals decode_scl_llr on
 * (double)llr[i]. Host-pointer and device-resident forms. */
int polar_decode_scl_llr_batch_f32(polar_code_t *h, const float *llr, long B, int L, uint8_t *out);
int polar_decode_scl_llr_batch_dev_f32(polar_code_t *h, const float *d_llr, long B, int L, uint8_t *d_out,
                                       double *d_pm, void *stream);
/* Pre-size the handle's device scratch for decodes of up to B codewords at list sizes 1 .. L (runs one decode per kernel
 * family on generated inputs — the list-size-1 kernel, the 2-lane groups, every power-of-two lane group up to L, with and
 * without d_pm — and waits for them). The device-resident entry points grow their scratch on demand — a hipFree/hipMalloc,
 * i.e. an implicit device synchronisation, whenever B or L exceeds anything seen before; after polar_reserve they do not
 * allocate for calls of at most B codewords and list size at most L, double or float LLRs, under the handle's current mode and
 * tuning (tests/test_gpu_parity.py asserts it on the allocation counter, polar_debug_get "allocs").
 * d_llr needs the natural alignment of its element type; rows that start 16-byte aligned (any hipMalloc'ed batch) let the
 * list-size-1 kernel read them in place, other pointers are decoded through a converted copy (same results). */
int polar_reserve(polar_code_t *h, long B, int L);
/* same, recording two hipEvent_t (may be NULL) on `stream` immediately around the launch of the
 * dominant kernel (scl_decode_llr_kernel), i.e. after the small all-frozen-prefix kernel — for
 * bench.py's roofline line */
int polar_decode_scl_llr_batch_dev_ev(polar_code_t *h, const double *d_llr, long B, int L, uint8_t *d_out,
                                      double *d_pm, void *stream, void *ev_start, void *ev_stop);

/* ---- PolarCode::decode_scl_p1 (PolarCode.cpp:110-128; PolarCode.m:299-310) ---- */
int polar_decode_scl_p1(polar_code_t *h, const double *p1 /*[N]*/, const double *p0 /*[N]*/, int L, uint8_t *out /*[K]*/);
int polar_decode_scl_p1_batch(polar_code_t *h, const double *p1, const double *p0, long B, int L, uint8_t *out);

/* ---- PolarM decode_sc_p1 (PolarCode.m:290-295, 870-895): SC on p1 = P(bit = 1) ----
 * out are doubles, as MATLAB returns them: 0 / 1, and 0.5 where a leaf probability is exactly 0.5
 * (sign(0) = 0 at PolarCode.m:873). */
int polar_decode_sc_p1(polar_code_t *h, const double *p1 /*[N]*/, double *out /*[K]*/);
int polar_decode_sc_p1_batch(polar_code_t *h, const double *p1, long B, double *out);

/* ---- synthetic BPSK/AWGN workload (include/polar_synth.h), generated on the device ----
 * trials [trial0, trial0+B): info bits (block = trial/100), encode, BPSK, AWGN, LLR with
 * the arithmetic of PolarCode.cpp:715,744-752; `s` = polar_snr_sqrt_linear(h, EbN0_dB).
 * d_info (optional) receives the transmitted info bits [B*K]. */
double polar_snr_sqrt_linear(const polar_code_t *h, double ebno_db);   /* PolarCode.cpp:744-745 */
int polar_synth_llr_dev(polar_code_t *h, uint64_t seed, uint64_t trial0, long B, double s,
                        double *d_llr, uint8_t *d_info, void *stream);
/* compare decoded vs sent info bits on the device: *d_err_count += #codewords that differ */
int polar_count_errors_dev(polar_code_t *h, const uint8_t *d_a, const uint8_t *d_b, long B,
                           unsigned long long *d_err_count, void *stream);

/* ---- PolarCode::get_bler_quick (PolarCode.cpp:658-785; PolarCode.m:781-850) ----
 * Batched Monte-Carlo on the synthetic workload. Semantics of the reference kept per trial
 * (one noise vector shared by every (L, Eb/N0); ascending Eb/N0 with "decoded at a lower
 * Eb/N0 => counted, not simulated", :728-742); the early stop `num_err > max_err` (:725)
 * is evaluated between rounds of `batch` trials (batch = 1 reproduces the reference's per-run granularity).
 * batch = 0 (the default of the host mirrors) picks the rounds itself: max(256, 2 max_err) trials first, then every
 * round as large as all rounds before it together (at most 262144) — a point overshoots the reference's stopping
 * time by less than 2x, and long sweeps still reach full-size launches. Reference defaults: max_runs = 1000,
 * max_err = 100 (:661-662); PolarM: 500 / 50 (PolarCode.m:788-789).
 * A round runs stream-ordered on the device (alive lists compacted there, PolarCode.cpp:728-742); the host reads
 * the 2 n_L n_e counters once per round. */
int polar_get_bler_quick(polar_code_t *h, const double *ebno, int n_e, const uint8_t *L, int n_L,
                         long max_runs, long max_err, uint64_t seed, long batch,
                         double *bler_out /*[n_L*n_e]*/);
/* PolarM's second output (PolarCode.m:781, 839, 848): ber[i] = (differing info bits of the block errors) / num_run,
 * per run as the reference computes it (NOT divided by K). Layout [n_L][n_e] like bler. */
int polar_get_bler_quick_ber(polar_code_t *h, const double *ebno, int n_e, const uint8_t *L, int n_L,
                             long max_runs, long max_err, uint64_t seed, long batch,
                             double *bler_out /*[n_L*n_e]*/, double *ber_out /*[n_L*n_e]*/);
/* The same sweep sharded over `n_dev` GPUs of this node from ONE host process (the C++ / MATLAB hosts): device
 * devices[d] (NULL = 0..n_dev-1) simulates the trials d, d + n_dev, ... of every round on its own stream, with its
 * own copy of the code tables and scratch (owned by `h`); the round's counters are summed with one RCCL
 * ncclAllReduce(uint64, sum) over xGMI (bound at run time; a host-side sum when RCCL cannot be loaded, or with
 * POLAR_NO_RCCL set when the handle was created). Counter-based inputs make the counters independent of n_dev for a given
 * `batch` (the automatic rounds grow with the device count: 262144 trials per device). *used_rccl (optional) reports
 * which path summed the counters. ber_out may be NULL. A device may be listed once (the test build of the library has a hook
 * that lifts this so that one GPU can stand in for several: include/polar_amd_debug.h). One worker thread per device lives on
 * the handle between calls (created with the communicators). */
int polar_get_bler_quick_multi(polar_code_t *h, const int *devices, int n_dev, const double *ebno, int n_e,
                               const uint8_t *L, int n_L, long max_runs, long max_err, uint64_t seed, long batch,
                               double *bler_out, double *ber_out, int *used_rccl);

/* General form of the sweep: `constellation` 0 / POLAR_CONST_BPSK = BPSK over AWGN with the Eb/N0 axis of
 * PolarCode.cpp:744-753 (what the three entry points above simulate); POLAR_CONST_ASK{4,8,16}_GRAY (include/polar_synth.h) =
 * the ASK Gray + BICM front end of PolarM/Constellation.m with the SNR axis and fresh info bits every run
 * (PolarM/main_MC_CC_Comparison.m:44-119: BASELINE configuration 5, sharded over the GPUs of the node from one host
 * process). devices == NULL: 0..n_dev-1 (with n_dev == 1: the handle's own device). Optional outputs (may be NULL):
 * ber_out, the raw counters err_out / run_out [n_L*n_e] (block errors and simulated-or-counted runs per point: what the
 * estimates are made of, and what two runs are compared by), *rounds_out = rounds the call took, *used_rccl.
 * Rounds: `batch` trials over all devices, or (batch == 0) geometric up to 262144 trials PER DEVICE.
 * The rounds are pipelined on the device — a step decodes point 1 of the newest round together with the later points of the
 * rounds before it, one launch per list size — with exactly the counters, early stop and run counts of the round-after-round
 * loop; the counters are reduced once per step.
 * Failure handling: a device that fails before the step's collective keeps every device out of it; a device whose
 * collective enqueue fails makes every device abort its communicator before it waits; a step that exceeds the watchdog
 * (1800 s; include/polar_amd_debug.h "multi_timeout_s") is ended in three bounded stages ("multi_grace_s", 10 s each): the workers are
 * signalled and abort their own communicators, what is left is aborted from the calling thread, and a worker that still does
 * not answer is given up — the call returns, the handle accepts no further get_bler_quick* calls and polar_destroy frees
 * nothing of it. Otherwise the call returns POLAR_E_DEVICE and the next call rebuilds the communicators. */
int polar_get_bler_quick_multi_ex(polar_code_t *h, int constellation, const int *devices, int n_dev, const double *axis, int n_e,
                                  const uint8_t *L, int n_L, long max_runs, long max_err, uint64_t seed, long batch,
                                  double *bler_out, double *ber_out, uint64_t *err_out, uint64_t *run_out, long *rounds_out,
                                  int *used_rccl);

/* The same sweep with the trials shared by `world` PROCESSES (one per GPU, as a process-group or MPI launcher starts them): this process is
 * `rank`, its handle's device simulates the trials rank, rank + world, ... of every round, and after every step `reduce` is
 * called — collectively, on every rank, the same number of times — to SUM the n uint64 counters in place over the ranks
 * (e.g. an all-reduce of the process group; return non-zero to fail the call). A rank whose own step failed (launch error,
 * watchdog) still makes this call once, with a failure flag in the last counter: every rank then returns POLAR_E_DEVICE from
 * the same step and none is left waiting in the collective. Counters, estimates and rounds are those of
 * polar_get_bler_quick_multi_ex with world devices. polar_amd/montecarlo.py drives it with its process group's all-reduce. */
typedef int (*polar_reduce_fn)(void *user, uint64_t *counters, int n);
int polar_get_bler_quick_rank(polar_code_t *h, int constellation, int rank, int world, polar_reduce_fn reduce, void *user,
                              const double *axis, int n_e, const uint8_t *L, int n_L, long max_runs, long max_err, uint64_t seed,
                              long batch, double *bler_out, double *ber_out, uint64_t *err_out, uint64_t *run_out, long *rounds_out);

/* step-wise Monte-Carlo for multi-GPU drivers: simulate trials {t0 + i*stride : i < T} for
 * every enabled (L, Eb/N0) point and ADD to err/run (host uint64 [n_L*n_e]). */
int polar_mc_batch(polar_code_t *h, uint64_t seed, uint64_t t0, long T, long stride,
                   const double *ebno, int n_e, const uint8_t *L, int n_L,
                   const uint8_t *enabled /*[n_L*n_e]*/, uint64_t *err, uint64_t *run);
/* same, also accumulating the differing info bits of the block errors (PolarM's num_bit_err, PolarCode.m:840) */
int polar_mc_batch_ber(polar_code_t *h, uint64_t seed, uint64_t t0, long T, long stride,
                       const double *ebno, int n_e, const uint8_t *L, int n_L,
                       const uint8_t *enabled /*[n_L*n_e]*/, uint64_t *err, uint64_t *bit_err, uint64_t *run);

/* ---- ASK Gray + BICM front end (PolarM/Constellation.m:84-93, 123-144; sweep conventions of
 * PolarM/main_MC_CC_Comparison.m:88-96) for the 16-ASK configuration: `constellation` is
 * POLAR_CONST_ASK{4,8,16}_GRAY (include/polar_synth.h), the sweep axis is the SNR in dB
 * (Eb/N0 = snr_db + 10log10(N/K) - 10log10(n_bits), main_MC_CC_Comparison.m:121), info bits are
 * fresh every run. Same counters/semantics as polar_mc_batch. */
int polar_synth_bicm_llr_dev(polar_code_t *h, int constellation, uint64_t seed, uint64_t trial0, long B,
                             double snr_db, double *d_llr, uint8_t *d_info, void *stream);
int polar_mc_batch_bicm(polar_code_t *h, int constellation, uint64_t seed, uint64_t t0, long T, long stride,
                        const double *snr_db, int n_s, const uint8_t *L, int n_L,
                        const uint8_t *enabled, uint64_t *err, uint64_t *run);

/* ---- Monte-Carlo code construction (PolarM/PolarCode.m:143-196 `monte_carlo`, receiver 'bicm',
 * with the genie-aided SC decoder `polar_decode_monte` :897-914). No handle: the result is what a
 * code is built FROM. For runs trial0 .. trial0+num_runs-1 (counter-based inputs, polar_synth.h):
 * N random message bits, polar transform, `constellation` (POLAR_CONST_BPSK or _ASK{4,8,16}_GRAY)
 * at the design SNR (sigma = sqrt(1/2) * 10^(-snr/20), n0 = sigma^2, :170), BICM p1, genie SC;
 * num_err[i] (host uint64 [2^n]) is INCREMENTED by the number of runs whose position i decided
 * wrongly — the table the reference writes to CodeConstructionData/MC_block_length_*.txt (:120-124)
 * and turns into a frozen set by a stable ascending sort (:126-135; polar_create_explicit /
 * PolarCode.from_counts). `batch` = runs per launch (0 = default). Disjoint trial ranges may be
 * summed across GPUs. */
int polar_mc_construction(int n, int constellation, double design_snr_db, uint64_t seed, uint64_t trial0,
                          long num_runs, long batch, uint64_t *num_err);

/* tuning knobs (0 = default): waves resident per CU and LDS-resident layer exponent */
int polar_set_tuning(polar_code_t *h, int waves_per_cu, int lds_log);
/* node arithmetic of decode_scl_llr: 0 = automatic (exp-domain kernel for list sizes >= 3, LLR-domain kernel
 * below), 1 = LLR-domain kernel only (table-driven exp/log1p f-node, the round-1 path), 2 = exp-domain kernel
 * (f-node = one division; codewords it cannot decide safely are flagged on the device and decoded again by the
 * LLR-domain kernel in the same call).
 * Decoded bits are the reference's in every mode for codes a construction produces for its channel (every BASELINE
 * configuration, the golden vectors, the fuzz slice of the -m gpu suite). Where unfrozen leaves lie in the worst synthetic
 * channels (explicit tables, rates near 1) the reference itself decides on the rounding noise of glibc's exp/log, which no
 * other arithmetic reproduces bit for bit: the handle marks such leaves at creation (BEC(1/2) capacity below 1e-3) and every
 * codeword in which one of them comes out below 1e-8 is decoded by the LLR-domain kernel whatever the mode, so automatic
 * mode is never worse there than mode 1 (HISTORY.md "Where bit-exactness ends"; tests/test_gpu_fuzz.py). With list sizes
 * below 3 mode 2 falls back to the LLR-domain kernel (the exp-domain kernels exist for groups of 4 lanes and more).
 * Environment overrides (measurement and tests only) are read ONCE, when a handle is created, and validated — no entry
 * point calls getenv afterwards: POLAR_MODE=<0|1|2> replaces the handle's mode (any other value: creation fails);
 * POLAR_SC_NO_FOLD=1 makes the list-size-1 kernel decode a permuted, converted copy of the batch (its round-2 front pass)
 * instead of reading the caller's rows in place; POLAR_NO_TABLES=1, POLAR_NO_RCCL=1, POLAR_FORCE_RCCL=1. Results do not
 * depend on any of them. */
int polar_set_mode(polar_code_t *h, int mode);
/* how many unfrozen leaves the handle classified as weak at creation (BEC(1/2) capacity below 1e-3; see above). Codes with
 * weak leaves are accepted, but their decoded bits are the reference's only as far as the LLR-domain kernel reproduces
 * glibc's rounding noise: polar_create_explicit() reports POLAR_W_WEAK_LEAVES (a positive, non-error status; the handle is
 * valid) and this function the count. */
int polar_get_weak_leaves(const polar_code_t *h);
/* Measurement knobs and test hooks (polar_debug_set / polar_debug_get / ...) are NOT part of this interface: they are declared
 * in include/polar_amd_debug.h, and the fault-injection hooks among them exist only in the test build of the library. */

#ifdef __cplusplus
}
#endif
#endif /* POLAR_AMD_H */
