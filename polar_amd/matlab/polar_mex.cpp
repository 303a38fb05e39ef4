// polar_mex.cpp — MEX gateway: command string + uint64 handle -> C-ABI (include/polar_amd.h).
// Build (needs MATLAB, not available in the build image — source delivered):
//   mex -I<repo>/include polar_mex.cpp -L<repo>/polar_amd -lpolar_amd
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "mex.h"
#include "polar_amd.h"
#include "polar_mex_layout.h"

static void check(int rc) {
    if (rc < 0) mexErrMsgIdAndTxt("polar_amd:error", "%s", polar_last_error());
}
static std::set<polar_code_t *> g_handles;      // handles this gateway created and has not destroyed (MATLAB calls MEX files from one thread)
static mxArray *new_handle(polar_code_t *h) {
    g_handles.insert(h);
    mexLock();
    mxArray *a = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t *)mxGetData(a) = (uint64_t)(uintptr_t)h;
    return a;
}
static void need(int nrhs, int k, const char *usage) {
    if (nrhs < k) mexErrMsgIdAndTxt("polar_amd:usage", "usage: polar_mex(%s)", usage);
}
static polar_code_t *H(const mxArray *a) {
    if (mxGetClassID(a) != mxUINT64_CLASS || mxGetNumberOfElements(a) != 1) mexErrMsgIdAndTxt("polar_amd:handle", "the handle must be the uint64 scalar 'create' returned");
    polar_code_t *h = (polar_code_t *)(uintptr_t)(*(uint64_t *)mxGetData(a));
    // (a MATLAB value outlives the object it names: a handle that was destroyed, or never was one, is refused instead of dereferenced)
    if (!g_handles.count(h)) mexErrMsgIdAndTxt("polar_amd:handle", "not a live polar_mex handle (destroyed, or from another MEX session)");
    return h;
}
static void want_class(const mxArray *a, mxClassID c, size_t min_elems, const char *what) {
    if (mxGetClassID(a) != c || mxGetNumberOfElements(a) < min_elems) mexErrMsgIdAndTxt("polar_amd:type", "%s: wrong class or too few elements", what);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("polar_amd:usage", "polar_mex(cmd, ...)");
    char cmd[64];
    mxGetString(prhs[0], cmd, sizeof cmd);
    std::string c(cmd);
    if (c == "create") {
        need(nrhs, 5, "'create', n, K, design_epsilon, crc_size");
        polar_code_t *h = nullptr;
        check(polar_create((int)mxGetScalar(prhs[1]), (int)mxGetScalar(prhs[2]), mxGetScalar(prhs[3]),
                           (int)mxGetScalar(prhs[4]), &h));
        plhs[0] = new_handle(h);
        return;
    }
    if (c == "create_explicit") {
        // polar_mex('create_explicit', n, K, crc, frozen(uint8 1xN), order(uint16 1xN, 0-based), crc_matrix(uint8 crc x K))
        need(nrhs, 6, "'create_explicit', n, K, crc, frozen, order0, crc_matrix");
        const int n_ = (int)mxGetScalar(prhs[1]), K_ = (int)mxGetScalar(prhs[2]), crc_ = (int)mxGetScalar(prhs[3]);
        if (n_ < 1 || n_ > POLAR_MAX_N_LOG2 || K_ < 1 || crc_ < 0) mexErrMsgIdAndTxt("polar_amd:size", "bad n / K / crc");
        want_class(prhs[4], mxUINT8_CLASS, (size_t)1 << n_, "frozen (uint8, 1 x N)");
        want_class(prhs[5], mxUINT16_CLASS, (size_t)1 << n_, "order (uint16, 1 x N, 0-based)");
        std::vector<uint8_t> m((size_t)crc_ * K_);
        if (crc_ > 0) {
            need(nrhs, 7, "'create_explicit', n, K, crc, frozen, order0, crc_matrix");
            want_class(prhs[6], mxUINT8_CLASS, (size_t)crc_ * K_, "crc_matrix (uint8, crc x K)");
            const uint8_t *d = (const uint8_t *)mxGetData(prhs[6]);           // column-major crc x K
            for (int i = 0; i < crc_; ++i)
                for (int j = 0; j < K_; ++j) m[(size_t)i * K_ + j] = d[(size_t)j * crc_ + i];
        }
        polar_code_t *h = nullptr;
        check(polar_create_explicit(n_, K_, crc_, (const uint8_t *)mxGetData(prhs[4]), (const uint16_t *)mxGetData(prhs[5]),
                                    crc_ > 0 ? m.data() : nullptr, &h));
        plhs[0] = new_handle(h);
        return;
    }
    if (c == "mc_construction") {
        // counts = polar_mex('mc_construction', n, constellation_id, design_snr_db, seed, num_runs [, trial0])  (PolarCode.m:143-196)
        // trial0 (default 0): first trial of the range — disjoint ranges (parallel workers, several GPUs) are summed by the caller
        need(nrhs, 6, "'mc_construction', n, constellation_id, design_snr_db, seed, num_runs");
        const int n_ = (int)mxGetScalar(prhs[1]);
        if (n_ < 1 || n_ > POLAR_MAX_N_LOG2) mexErrMsgIdAndTxt("polar_amd:size", "bad n");
        std::vector<uint64_t> cnt((size_t)1 << n_, 0);
        check(polar_mc_construction(n_, (int)mxGetScalar(prhs[2]), mxGetScalar(prhs[3]), (uint64_t)mxGetScalar(prhs[4]),
                                    nrhs > 6 ? (uint64_t)mxGetScalar(prhs[6]) : 0, (long)mxGetScalar(prhs[5]), 0, cnt.data()));
        plhs[0] = mxCreateDoubleMatrix((mwSize)cnt.size(), 1, mxREAL);
        for (size_t i = 0; i < cnt.size(); ++i) mxGetPr(plhs[0])[i] = (double)cnt[i];
        return;
    }
    if (c == "monte_carlo_design") {
        // [h, frozen, order0, bler_est] = polar_mex('monte_carlo_design', n, K, crc, crc_matrix, constellation_id,
        //                                           design_snr_db, num_runs, seed, table_file)
        // The whole design step in one call: per-channel error counts from `table_file` when it exists (the
        // reference's cache format, one '%d ' per line), else from the GPU (polar_mc_construction) and written
        // there; reliability order = stable ascending sort of the counts; explicit-table handle.
        need(nrhs, 9, "'monte_carlo_design', n, K, crc, crc_matrix, constellation_id, design_snr_db, num_runs, seed [, table_file]");
        const int n_ = (int)mxGetScalar(prhs[1]), K_ = (int)mxGetScalar(prhs[2]), crc_ = (int)mxGetScalar(prhs[3]);
        if (n_ < 1 || n_ > POLAR_MAX_N_LOG2 || K_ < 1 || crc_ < 0 || (size_t)(K_ + crc_) > ((size_t)1 << n_)) mexErrMsgIdAndTxt("polar_amd:size", "bad n / K / crc");
        if (crc_ > 0) want_class(prhs[4], mxUINT8_CLASS, (size_t)crc_ * K_, "crc_matrix (uint8, crc x K)");
        const size_t N_ = (size_t)1 << n_;
        char path[1024] = "";
        if (nrhs > 9 && mxIsChar(prhs[9])) mxGetString(prhs[9], path, sizeof path);
        std::vector<uint64_t> cnt(N_, 0);
        bool have = false;
        if (path[0]) {
            if (FILE *f = fopen(path, "r")) {
                size_t k = 0;
                double v;
                while (k < N_ && fscanf(f, "%lf", &v) == 1) cnt[k++] = (uint64_t)v;
                fclose(f);
                have = (k == N_);
            }
        }
        const long runs = (long)mxGetScalar(prhs[7]);
        if (!have) {
            check(polar_mc_construction(n_, (int)mxGetScalar(prhs[5]), mxGetScalar(prhs[6]), (uint64_t)mxGetScalar(prhs[8]), 0, runs, 0, cnt.data()));
            if (path[0]) {
                if (FILE *f = fopen(path, "w")) {
                    for (size_t i = 0; i < N_; ++i) fprintf(f, "%llu \n", (unsigned long long)cnt[i]);
                    fclose(f);
                }
            }
        }
        std::vector<uint16_t> order(N_);
        for (size_t i = 0; i < N_; ++i) order[i] = (uint16_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint16_t a, uint16_t b) { return cnt[a] < cnt[b]; });
        std::vector<uint8_t> frozen(N_, 1), m((size_t)crc_ * K_);
        double est = 0;
        for (int i = 0; i < K_ + crc_; ++i) { frozen[order[i]] = 0; est += (double)cnt[order[i]]; }
        if (crc_ > 0) {
            const uint8_t *d = (const uint8_t *)mxGetData(prhs[4]);           // column-major crc x K
            for (int i = 0; i < crc_; ++i)
                for (int j = 0; j < K_; ++j) m[(size_t)i * K_ + j] = d[(size_t)j * crc_ + i];
        }
        polar_code_t *h = nullptr;
        check(polar_create_explicit(n_, K_, crc_, frozen.data(), order.data(), crc_ > 0 ? m.data() : nullptr, &h));
        plhs[0] = new_handle(h);
        if (nlhs > 1) { plhs[1] = mxCreateNumericMatrix(1, N_, mxUINT8_CLASS, mxREAL); memcpy(mxGetData(plhs[1]), frozen.data(), N_); }
        if (nlhs > 2) { plhs[2] = mxCreateNumericMatrix(1, N_, mxUINT16_CLASS, mxREAL); memcpy(mxGetData(plhs[2]), order.data(), 2 * N_); }
        if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(est / (double)runs);
        return;
    }
    need(nrhs, 2, "cmd, handle, ...");
    polar_code_t *h = H(prhs[1]);
    int n, N, K, crc;
    check(polar_get_params(h, &n, &N, &K, &crc));
    if (c == "destroy") {
        polar_destroy(h);
        g_handles.erase(h);
        mexUnlock();
    } else if (c == "tables") {
        if (nlhs < 3) mexErrMsgIdAndTxt("polar_amd:usage", "usage: [frozen, order0, crc_matrix] = polar_mex('tables', h)");
        plhs[0] = mxCreateNumericMatrix(1, N, mxUINT8_CLASS, mxREAL);
        check(polar_get_frozen(h, (uint8_t *)mxGetData(plhs[0])));
        plhs[1] = mxCreateNumericMatrix(1, N, mxUINT16_CLASS, mxREAL);
        check(polar_get_order(h, (uint16_t *)mxGetData(plhs[1])));
        std::vector<uint8_t> m((size_t)crc * K);
        check(polar_get_crc_matrix(h, m.data()));
        plhs[2] = mxCreateNumericMatrix(crc, K, mxUINT8_CLASS, mxREAL);   // column-major
        uint8_t *d = (uint8_t *)mxGetData(plhs[2]);
        for (int i = 0; i < crc; ++i)
            for (int j = 0; j < K; ++j) d[(size_t)j * crc + i] = m[(size_t)i * K + j];
    } else if (c == "encode") {
        need(nrhs, 3, "'encode', h, info_bits(uint8 1 x K)");
        want_class(prhs[2], mxUINT8_CLASS, (size_t)K, "info_bits (uint8, 1 x K)");
        plhs[0] = mxCreateNumericMatrix(1, N, mxUINT8_CLASS, mxREAL);
        check(polar_encode(h, (const uint8_t *)mxGetData(prhs[2]), (uint8_t *)mxGetData(plhs[0])));
    } else if (c == "decode_scl_llr") {
        // u = polar_mex('decode_scl_llr', h, llr, list_size [, layout]): llr double or single; 1 x N, B x N (one codeword per row ->
        // u is B x K) or N x B (one codeword per COLUMN -> u is K x B: MATLAB's storage is then the library's, nothing is copied
        // on either side). layout: 'rows' | 'cols' says which; without it the shape decides, and a square N x N argument — where
        // it cannot — is an error rather than a guess. Batches from 32 MiB on are pipelined inside the library (pinned staging,
        // copy stream, decode lanes).
        need(nrhs, 4, "'decode_scl_llr', h, llr, list_size [, 'rows' | 'cols']");
        char want = 0;
        if (nrhs > 4) {
            char lay_s[16] = "";
            if (!mxIsChar(prhs[4]) || mxGetString(prhs[4], lay_s, sizeof lay_s)) mexErrMsgIdAndTxt("polar_amd:layout", "layout must be 'rows' or 'cols'");
            if (std::string(lay_s) == "rows") want = 'r';
            else if (std::string(lay_s) == "cols") want = 'c';
            else mexErrMsgIdAndTxt("polar_amd:layout", "layout must be 'rows' or 'cols' (got '%s')", lay_s);
        }
        size_t B = 0;
        bool ambiguous = false;
        const char lay = polar_mex::batch_layout(mxGetM(prhs[2]), mxGetN(prhs[2]), (size_t)N, want, &B, &ambiguous);
        if (!lay) mexErrMsgIdAndTxt("polar_amd:size", "llr must be 1 x N, B x N or N x B (N = %d)%s", N, want ? ", in the layout named" : "");
        if (ambiguous) mexErrMsgIdAndTxt("polar_amd:layout", "a %d x %d llr can hold its codewords as rows or as columns: pass 'rows' or 'cols'", N, N);
        const bool f32 = mxGetClassID(prhs[2]) == mxSINGLE_CLASS;
        if (!f32 && mxGetClassID(prhs[2]) != mxDOUBLE_CLASS) mexErrMsgIdAndTxt("polar_amd:type", "llr must be double or single");
        const int L = (int)mxGetScalar(prhs[3]);
        const void *src = mxGetData(prhs[2]);
        std::vector<double> t64;
        std::vector<float> t32;
        if (lay == 'r' && B > 1) {                       // gather the rows (blocked, multi-threaded)
            if (f32) { t32.resize(B * N); polar_mex::rows_from_colmajor((const float *)src, B, (size_t)N, t32.data()); src = t32.data(); }
            else { t64.resize(B * N); polar_mex::rows_from_colmajor((const double *)src, B, (size_t)N, t64.data()); src = t64.data(); }
        }
        if (lay == 'c') {
            plhs[0] = mxCreateNumericMatrix(K, B, mxUINT8_CLASS, mxREAL);
            uint8_t *d = (uint8_t *)mxGetData(plhs[0]);
            check(f32 ? polar_decode_scl_llr_batch_f32(h, (const float *)src, (long)B, L, d) : polar_decode_scl_llr_batch(h, (const double *)src, (long)B, L, d));
        } else {
            std::vector<uint8_t> out(B * K);
            check(f32 ? polar_decode_scl_llr_batch_f32(h, (const float *)src, (long)B, L, out.data()) : polar_decode_scl_llr_batch(h, (const double *)src, (long)B, L, out.data()));
            plhs[0] = mxCreateNumericMatrix(B, K, mxUINT8_CLASS, mxREAL);
            polar_mex::colmajor_from_rows(out.data(), B, (size_t)K, (uint8_t *)mxGetData(plhs[0]));
        }
    } else if (c == "decode_scl_p1") {
        need(nrhs, 5, "'decode_scl_p1', h, p1, p0, list_size");
        want_class(prhs[2], mxDOUBLE_CLASS, (size_t)N, "p1 (double, 1 x N)");
        want_class(prhs[3], mxDOUBLE_CLASS, (size_t)N, "p0 (double, 1 x N)");
        plhs[0] = mxCreateNumericMatrix(1, K, mxUINT8_CLASS, mxREAL);
        check(polar_decode_scl_p1(h, mxGetPr(prhs[2]), mxGetPr(prhs[3]), (int)mxGetScalar(prhs[4]),
                                  (uint8_t *)mxGetData(plhs[0])));
    } else if (c == "decode_sc_p1") {
        need(nrhs, 3, "'decode_sc_p1', h, p1");
        want_class(prhs[2], mxDOUBLE_CLASS, (size_t)N, "p1 (double, 1 x N)");
        plhs[0] = mxCreateDoubleMatrix(1, K, mxREAL);
        check(polar_decode_sc_p1(h, mxGetPr(prhs[2]), mxGetPr(plhs[0])));
    } else if (c == "get_bler_quick") {
        // [bler, ber] = polar_mex('get_bler_quick', h, axis(1 x n_e), L(uint8 1 x n_L), max_runs, max_err, seed [, devices(int32) [, constellation_id]])
        // both outputs n_L x n_e (PolarC layout); with a device list the trials are sharded over those GPUs; constellation_id
        // 0 = BPSK with the Eb/N0 axis, POLAR_CONST_ASK*_GRAY = the BICM sweep with the SNR axis (main_MC_CC_Comparison.m:44-119)
        need(nrhs, 7, "'get_bler_quick', h, axis, list_sizes(uint8), max_runs, max_err, seed [, devices(int32) [, constellation_id]]");
        want_class(prhs[2], mxDOUBLE_CLASS, 1, "axis (double)");
        want_class(prhs[3], mxUINT8_CLASS, 1, "list sizes (uint8)");
        if (nrhs > 7 && mxGetNumberOfElements(prhs[7]) > 0) want_class(prhs[7], mxINT32_CLASS, 1, "devices (int32)");
        int n_e = (int)mxGetNumberOfElements(prhs[2]), n_L = (int)mxGetNumberOfElements(prhs[3]);
        long max_runs = (long)mxGetScalar(prhs[4]), max_err = (long)mxGetScalar(prhs[5]);
        uint64_t seed = (uint64_t)mxGetScalar(prhs[6]);
        std::vector<double> b((size_t)n_e * n_L), e((size_t)n_e * n_L);
        const bool have_devs = nrhs > 7 && mxGetNumberOfElements(prhs[7]) > 0;
        const int constellation = nrhs > 8 ? (int)mxGetScalar(prhs[8]) : 0;
        check(polar_get_bler_quick_multi_ex(h, constellation, have_devs ? (const int *)mxGetData(prhs[7]) : nullptr,
                                            have_devs ? (int)mxGetNumberOfElements(prhs[7]) : 1, mxGetPr(prhs[2]), n_e,
                                            (const uint8_t *)mxGetData(prhs[3]), n_L, max_runs, max_err, seed, 0, b.data(), e.data(),
                                            nullptr, nullptr, nullptr, nullptr));
        plhs[0] = mxCreateDoubleMatrix(n_L, n_e, mxREAL);
        double *d = mxGetPr(plhs[0]);
        for (int l = 0; l < n_L; ++l) for (int i = 0; i < n_e; ++i) d[(size_t)i * n_L + l] = b[(size_t)l * n_e + i];
        if (nlhs > 1) {
            plhs[1] = mxCreateDoubleMatrix(n_L, n_e, mxREAL);
            double *d2 = mxGetPr(plhs[1]);
            for (int l = 0; l < n_L; ++l) for (int i = 0; i < n_e; ++i) d2[(size_t)i * n_L + l] = e[(size_t)l * n_e + i];
        }
    } else {
        mexErrMsgIdAndTxt("polar_amd:cmd", "unknown command %s", cmd);
    }
}
