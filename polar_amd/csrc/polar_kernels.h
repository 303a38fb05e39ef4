// polar_kernels.h — launch interface between the C-ABI host code and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

struct PolarDecodeParams {
    int n, N, K, crc, L;
    int W;                       // 32-bit words of decision history = ceil((K+crc)/32)
    int prefix_q, prefix_len;    // all-frozen prefix handled cooperatively: block size Q (0 = off), leaves Pe
    long B;                      // codewords
    const double *llr;           // [B][N] device (LLR mode: llr; probability mode: p1); floats when llr_f32
    int llr_f32;                 // channel LLRs are float[B][N] (widened in the load)
    const double *p0;            // [B][N] device (probability mode only)
    uint8_t *out;                // [B][K] device
    double *pm_out;              // [B] device or nullptr
    const uint8_t *frozen;       // [N] device
    const uint32_t *ctl;         // [N] device: per-leaf control word of the LLR kernel = frozen | zb << 1 | weak << 8, zb = log2 size (0, 2 or 3) of the all-frozen aligned block starting at this leaf, weak = unfrozen leaf in one of the worst channels (scalar-loaded)
    unsigned int *work;          // device counter (zeroed before the launch): dynamic hand-out of codeword groups after a wave's first one
    const uint16_t *info_rank;   // [K+crc] device: rank of order[beta] among the unfrozen positions
    const uint32_t *crc_mask;    // [crc][W] device: parity masks over unfrozen ranks (check bit included)
    const double *tabs;          // [322] device: T[64] = 2^(-j/64), RC[129] = 1/(1+j/128), LC[129] = log(1+j/128)
    const double *pre;           // [B][N - Q + 1] device: prefix_kernel output (metric + node-0 f-chain), nullptr if off
    double *llr_scr;             // per-wave scratch: [grid][N - 2*SL][64]
    uint32_t *c_scr;             // per-wave scratch: [grid][2][N/32 - 2][64]
    uint32_t *hist_scr;          // per-wave scratch: [grid][W][64]
    // exp-domain fast path + fallback pass
    uint8_t *flags;              // [B] device (ED kernels): 1 = decode this codeword again with the LLR-domain kernel
    const uint32_t *cw_list;     // work list of codeword indices (fallback pass), nullptr = 0..B-1
    const unsigned int *cw_count;// device: number of entries of cw_list (read by the kernel), nullptr = B
    const unsigned int *n_dev;   // device: only the first min(B, *n_dev) codewords exist (Monte-Carlo alive lists), nullptr = B
    double *tab_scr;             // table mode (GS = 32, exp-domain): per-wave [grid][2][3N] layer-1/-2 value tables, nullptr = off
    uint32_t *var_scr;           //   per-wave [grid][N/32][64] variant nibbles of the paths
};

size_t polar_decode_lds_bytes(int lds_log, int pipe);
int polar_decode_waves_per_block(int pipe);
hipError_t polar_launch_prefix(const PolarDecodeParams &p, bool ed, double *ech_out, hipStream_t st);
hipError_t polar_launch_prefix_ed0(const PolarDecodeParams &p, double *ech_out, hipStream_t st);
hipError_t polar_launch_prefix_ed1(const PolarDecodeParams &p, double *ech_out, hipStream_t st);
int polar_prefix_is_staged(int N);
hipError_t polar_launch_decode_llr_ed0(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, hipStream_t st);
hipError_t polar_launch_decode_llr_ed1(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, hipStream_t st);
// one codeword per wave, state in LDS (the latency form; list sizes 2 .. 8 while polar_decode_lat_lds_bytes() fits 160 KiB)
hipError_t polar_launch_decode_lat(const PolarDecodeParams &p, int gs, bool ed, int blocks, hipStream_t st);
size_t polar_decode_lat_lds_bytes(int N, int gs, int W);
hipError_t polar_launch_decode_llr(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, bool ed, hipStream_t st);
hipError_t polar_launch_ed_front(const void *llr, int llr_f32, double *ech, uint8_t *flags, const double *tabs, int N, long B, const unsigned *n_dev, hipStream_t st);
hipError_t polar_launch_ed_collect(const uint8_t *flags, long B, const unsigned *n_dev, uint32_t *list, unsigned *count, hipStream_t st);

hipError_t polar_launch_decode_p1(const PolarDecodeParams &p, int gs, int grid, hipStream_t st);

struct PolarScP1Params {
    int n, N, K;
    long B;
    const double *p1;            // [B][N] device
    double *out;                 // [B][K] device (doubles, as MATLAB: 0.5 possible)
    const uint8_t *frozen;       // [N]
    const uint16_t *order;       // [N]
    double *scr;                 // per-wave scratch [grid][4*N][64]
};
hipError_t polar_launch_sc_p1(const PolarScP1Params &p, int grid, hipStream_t st);
// small batches: one codeword per wave, state in LDS (polar_sc_p1_lat_lds_bytes(N) must fit the device's LDS)
hipError_t polar_launch_sc_p1_lat(const PolarScP1Params &p, int grid, hipStream_t st);
size_t polar_sc_p1_lat_lds_bytes(int N);

// list size 1: pruned successive cancellation (polar_kernels_sc.hip)
struct PolarScParams {
    int n, N, K;
    long B;
    const double *ech_t;         // [B][N] device: channel values, stored form, kernel element order (sc8_front_kernel); unused when `llr` is set
    const void *llr;             // nullptr, or [B][N] device: the caller's rows (double, or float when llr_f32), read IN PLACE by the two
    int llr_f32;                 //   visits of the top layer (no front pass): polar_sc8_can_fold() says for which schedules
    uint8_t *out;                // [B][K] device
    const uint32_t *ops;         // [n_ops] device: schedule words = type | log2(S) << 3 | first leaf << 8
    int n_ops;                   //   type 0 F, 1 G, 3 all-unfrozen, 4 combine, 6 all-frozen bound
    const uint16_t *order;       // [N] device (the first K entries are read)
    const double *tabs;          // [322] device
    double *a_scr;               // per-wave scratch: the layers larger than the LDS-resident ones, polar_sc8_scratch_doubles_per_wave()
    unsigned int *flag_words;    // [ceil(B/32)] device, bit = codeword to be decoded again by the general kernel
    uint8_t *flag_bytes;         // nullptr, or [B] (sc_lat_kernel only): the same flag as a byte per codeword, WRITTEN for every codeword
                                 //   (the zero-copy host path reads it from pinned memory instead of copying the flag words back)
    unsigned int *work;          // device counter (zeroed before the launch) or nullptr
    const unsigned int *n_dev;   // device: only the first min(B, *n_dev) codewords exist, nullptr = B
};
// eight lanes per codeword (channel values permuted per codeword [B][N])
size_t polar_sc8_lds_bytes(int N);
int polar_sc8_waves_per_block();
int polar_sc8_waves_per_cu(int N);
size_t polar_sc8_scratch_doubles_per_wave(int N);
int polar_sc8_fold_min_log();            // smallest log2(block length) whose top-layer visits can read the caller's rows in place
int polar_sc8_min_global_log();          // log2 of the smallest HBM-resident layer of the list-size-1 kernel
hipError_t polar_launch_sc8_front(const void *llr, int llr_f32, double *ech_p, unsigned int *flag_words, const double *tabs,
                                  int n, long B, const unsigned *n_dev, hipStream_t st);
hipError_t polar_launch_sc8_decode(const PolarScParams &p, int grid_waves, hipStream_t st);
// one codeword per wave, whole state in LDS: the latency form for small batches (N <= 2^polar_sc_lat_max_log())
size_t polar_sc_lat_lds_bytes(int N, int n_ops);
int polar_sc_lat_max_log();
hipError_t polar_launch_sc_lat(const PolarScParams &p, int blocks, hipStream_t st);
hipError_t polar_launch_sc_collect(const unsigned int *flag_words, long B, const unsigned *n_dev, uint32_t *list, unsigned *count, hipStream_t st);

// Monte-Carlo code construction (polar_construct.hip)
struct PolarConstructParams {
    int n, N;
    long B;                      // runs in this batch
    uint64_t seed, trial0;       // run b uses trial index trial0 + b
    int constellation;           // POLAR_CONST_*
    double sigma, n0, cnorm;
    double *p1;                  // [B][N] device: P(bit = 1) per position
    uint32_t *info;              // [B][ceil(N/32)] device: packed message bits
    double *y_scr;               // per-wave scratch [grid][N][64]
    uint8_t *x_scr;              // per-wave scratch [grid][2*N][64]
    unsigned long long *num_err; // [N] device accumulators
};
hipError_t polar_launch_mc_front(const PolarConstructParams &p, int grid, hipStream_t st);
hipError_t polar_launch_mc_genie(const PolarConstructParams &p, int grid, hipStream_t st);

struct PolarEncodeParams {
    int n, N, K, crc;
    long B;
    const uint8_t *info;         // [B][K] device (encode) — unused by synth
    uint8_t *coded;              // [B][N] device (encode) — optional for synth
    const uint16_t *order;       // [N] device
    const uint8_t *crcm;         // [crc][K] device
    // synth
    uint64_t seed, trial0;
    long stride;                 // trial index = trial0 + b*stride, or sel[b] when sel != nullptr
    const uint64_t *sel;
    double s;
    int constellation;           // 0 = BPSK (s), else POLAR_CONST_* (sigma, n0, cnorm)
    double sigma, n0, cnorm;
    long info_block_div;         // info bits keyed by trial / info_block_div (100 = reference's refresh, 1 = every run)
    double *llr;                 // [B][N]
    uint8_t *info_out;           // [B][K] or nullptr
    const unsigned int *n_dev;   // device: only the first min(B, *n_dev) rows exist (Monte-Carlo alive lists), nullptr = B
};
hipError_t polar_launch_encode(const PolarEncodeParams &p, hipStream_t st);
hipError_t polar_launch_synth(const PolarEncodeParams &p, hipStream_t st);   // BPSK or ASK/BICM by p.constellation
hipError_t polar_launch_count_errors(const uint8_t *a, const uint8_t *b, long B, int K,
                                     unsigned long long *err, uint8_t *mismatch_flags, hipStream_t st);
// Monte-Carlo round on the device (PolarCode.cpp:728-742, 758-769): alive[i] = t0 + i*stride, *n = T
hipError_t polar_launch_mc_init_alive(uint64_t *alive, unsigned *n, uint64_t t0, long stride, long T, hipStream_t st);
// rows [0, min(B, *n_in)): block error iff decoded != sent; ctr[0] += block errors, ctr[1] += differing bits
// (PolarM/PolarCode.m:836-840); the trials in error are appended to alive_out / *n_out (the others were decoded
// correctly and are "counted, not simulated" at the higher Eb/N0 points)
hipError_t polar_launch_mc_count_compact(const uint8_t *decoded, const uint8_t *sent, long B, int K,
                                         const uint64_t *alive_in, const unsigned *n_in, uint64_t *alive_out, unsigned *n_out,
                                         unsigned long long *ctr, hipStream_t st);
