// polar_mex_layout.h — MATLAB (column-major) <-> the C-ABI's row-major, codeword-contiguous batches, for the MEX gateway.
// Pure C++ (no mex.h): tests/test_abi.py compiles and runs it against a naive loop.
//
// A batch of B codewords reaches the gateway either as N x B (one codeword per COLUMN: already codeword-contiguous in MATLAB's
// column-major storage — handed to the library as it is, no copy) or as B x N (one codeword per ROW, the layout of PolarM's row
// vectors stacked): then the rows are gathered by a blocked, multi-threaded transpose (a scalar strided loop moves 1 GiB in
// seconds; the pipelined decode of the same batch takes 20-70 ms).
#pragma once
#include <algorithm>
#include <cstddef>
#include <thread>
#include <vector>

namespace polar_mex {

// y[b * N + i] = x[i * B + b]   (x: B x N column-major, y: B rows of N)
template <typename T>
void rows_from_colmajor(const T *x, size_t B, size_t N, T *y, unsigned threads = 0) {
    constexpr size_t TB = 32;                                  // tile: 32 x 32 elements (8 KiB of doubles per side)
    if (threads == 0) threads = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
    const size_t nb = (B + TB - 1) / TB;
    if (B * N < ((size_t)1 << 20)) threads = 1;
    auto work = [&](size_t t0, size_t t1) {
        for (size_t tb = t0; tb < t1; ++tb) {
            const size_t b0 = tb * TB, b1 = std::min(B, b0 + TB);
            for (size_t i0 = 0; i0 < N; i0 += TB) {
                const size_t i1 = std::min(N, i0 + TB);
                for (size_t i = i0; i < i1; ++i)
                    for (size_t b = b0; b < b1; ++b) y[b * N + i] = x[i * B + b];
            }
        }
    };
    if (threads <= 1) { work(0, nb); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; ++t) th.emplace_back(work, nb * t / threads, nb * (t + 1) / threads);
    for (auto &q : th) q.join();
}
// x[i * B + b] = y[b * N + i]   (the inverse: B rows of N -> B x N column-major)
template <typename T>
void colmajor_from_rows(const T *y, size_t B, size_t N, T *x, unsigned threads = 0) {
    constexpr size_t TB = 32;
    if (threads == 0) threads = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
    const size_t nb = (B + TB - 1) / TB;
    if (B * N < ((size_t)1 << 20)) threads = 1;
    auto work = [&](size_t t0, size_t t1) {
        for (size_t tb = t0; tb < t1; ++tb) {
            const size_t b0 = tb * TB, b1 = std::min(B, b0 + TB);
            for (size_t i0 = 0; i0 < N; i0 += TB) {
                const size_t i1 = std::min(N, i0 + TB);
                for (size_t b = b0; b < b1; ++b)
                    for (size_t i = i0; i < i1; ++i) x[i * B + b] = y[b * N + i];
            }
        }
    };
    if (threads <= 1) { work(0, nb); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; ++t) th.emplace_back(work, nb * t / threads, nb * (t + 1) / threads);
    for (auto &q : th) q.join();
}
// how a 2-D argument of `rows` x `cols` holds codewords of length N: 'c' = one per column (N x B), 'r' = one per row (B x N;
// also a single 1 x N or N x 1 vector), 0 = neither. `want` ('r' / 'c' / 0) is the caller's explicit choice: the shape must
// then fit it. Without one the shape decides; a square N x N argument fits both — *ambiguous is set and the caller refuses it
// (round 5 read it as rows, silently: 2048 codewords stored as columns came back as wrong bits).
inline char batch_layout(size_t rows, size_t cols, size_t N, char want, size_t *B, bool *ambiguous) {
    *ambiguous = false;
    if (rows * cols == N && (rows == 1 || cols == 1)) { *B = 1; return want ? want : 'r'; }      // (one vector is one codeword either way)
    if (want == 'r') { if (cols != N) return 0; *B = rows; return 'r'; }
    if (want == 'c') { if (rows != N) return 0; *B = cols; return 'c'; }
    if (cols == N && rows == N) { *B = N; *ambiguous = true; return 'r'; }
    if (cols == N) { *B = rows; return 'r'; }
    if (rows == N) { *B = cols; return 'c'; }
    return 0;
}

}  // namespace polar_mex
