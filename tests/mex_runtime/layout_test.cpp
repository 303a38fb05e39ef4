// tests/mex_runtime/layout_test.cpp — the MEX gateway's layout helpers (polar_amd/matlab/polar_mex_layout.h) against a naive loop.
#include <cstdint>
#include <cstdio>
#include <vector>
#include "polar_mex_layout.h"
template <typename T>
static int check(size_t B, size_t N, unsigned threads) {
    std::vector<T> x(B * N), y(B * N, (T)0), z(B * N, (T)0);
    for (size_t i = 0; i < B * N; ++i) x[i] = (T)((i * 2654435761u) % 251);
    polar_mex::rows_from_colmajor(x.data(), B, N, y.data(), threads);
    for (size_t b = 0; b < B; ++b) for (size_t i = 0; i < N; ++i) if (y[b * N + i] != x[i * B + b]) return 1;
    polar_mex::colmajor_from_rows(y.data(), B, N, z.data(), threads);
    for (size_t i = 0; i < B * N; ++i) if (z[i] != x[i]) return 2;
    return 0;
}
int main() {
    int bad = 0;
    for (unsigned th : {1u, 3u, 0u})
        for (auto bn : {std::pair<size_t, size_t>{1, 8}, {3, 5}, {31, 33}, {32, 32}, {65, 127}, {1000, 256}, {700, 2048}}) {
            bad += check<double>(bn.first, bn.second, th); bad += check<float>(bn.first, bn.second, th); bad += check<uint8_t>(bn.first, bn.second, th);
        }
    size_t B = 0;
    bool amb = false;
    bad += polar_mex::batch_layout(1, 2048, 2048, 0, &B, &amb) != 'r' || B != 1 || amb;
    bad += polar_mex::batch_layout(2048, 1, 2048, 0, &B, &amb) != 'r' || B != 1 || amb;
    bad += polar_mex::batch_layout(7, 2048, 2048, 0, &B, &amb) != 'r' || B != 7 || amb;
    bad += polar_mex::batch_layout(2048, 7, 2048, 0, &B, &amb) != 'c' || B != 7 || amb;
    bad += polar_mex::batch_layout(2048, 2048, 2048, 0, &B, &amb) == 0 || !amb;            // square: the caller must say
    bad += polar_mex::batch_layout(2048, 2048, 2048, 'r', &B, &amb) != 'r' || B != 2048 || amb;
    bad += polar_mex::batch_layout(2048, 2048, 2048, 'c', &B, &amb) != 'c' || B != 2048 || amb;
    bad += polar_mex::batch_layout(7, 2048, 2048, 'c', &B, &amb) != 0;                     // named layout the shape does not fit
    bad += polar_mex::batch_layout(2048, 7, 2048, 'r', &B, &amb) != 0;
    bad += polar_mex::batch_layout(5, 6, 2048, 0, &B, &amb) != 0;
    std::printf("%s\n", bad ? "FAIL" : "ok");
    return bad ? 1 : 0;
}
