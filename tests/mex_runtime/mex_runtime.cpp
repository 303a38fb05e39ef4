// tests/mex_runtime/mex_runtime.cpp — the working mx / mex runtime behind tests/mex_runtime/mex.h, and the C entry points
// (fm_*) through which tests/fake_matlab.py builds arguments, calls the gateway's mexFunction and reads the results.
// Test infrastructure: never part of the product library.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <new>
#include <string>

#include "mex.h"

struct mxArray_tag {
    mxClassID cls;
    size_t m, n;
    void *data;        // column-major, m * n elements of elem_size(cls); NULL when empty
};

namespace {
std::atomic<long> g_live{0};      // arrays alive (leak check of the driver)
int g_lock = 0;                   // mexLock depth

size_t elem_size(mxClassID c) {
    switch (c) {
    case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
    case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxCHAR_CLASS: case mxINT16_CLASS: case mxUINT16_CLASS: return 2;
    case mxLOGICAL_CLASS: case mxINT8_CLASS: case mxUINT8_CLASS: return 1;
    default: return 0;
    }
}
struct MexError {
    std::string id, msg;
};
mxArray *make(mxClassID c, size_t m, size_t n) {
    const size_t es = elem_size(c);
    if (!es) throw MexError{"fake_mx:class", "unsupported class id"};
    mxArray *a = new mxArray_tag{c, m, n, nullptr};
    if (m * n) {
        a->data = calloc(m * n, es);          // MATLAB zero-initialises numeric arrays
        if (!a->data) { delete a; throw std::bad_alloc(); }
    }
    ++g_live;
    return a;
}
}  // namespace

extern "C" {

bool mxIsChar(const mxArray *a) { return a && a->cls == mxCHAR_CLASS; }
int mxGetString(const mxArray *a, char *buf, mwSize len) {
    if (!a || a->cls != mxCHAR_CLASS || !buf || len == 0) return 1;
    const size_t k = a->m * a->n;
    const mxChar *s = (const mxChar *)a->data;
    const size_t c = k < len - 1 ? k : len - 1;
    for (size_t i = 0; i < c; ++i) buf[i] = (char)s[i];
    buf[c] = 0;
    return k > len - 1 ? 1 : 0;
}
double mxGetScalar(const mxArray *a) {
    if (!a || !a->data || a->m * a->n == 0) throw MexError{"fake_mx:scalar", "mxGetScalar of an empty array"};
    switch (a->cls) {
    case mxDOUBLE_CLASS: return *(const double *)a->data;
    case mxSINGLE_CLASS: return *(const float *)a->data;
    case mxINT8_CLASS: return *(const int8_t *)a->data;
    case mxUINT8_CLASS: case mxLOGICAL_CLASS: return *(const uint8_t *)a->data;
    case mxINT16_CLASS: return *(const int16_t *)a->data;
    case mxUINT16_CLASS: case mxCHAR_CLASS: return *(const uint16_t *)a->data;
    case mxINT32_CLASS: return *(const int32_t *)a->data;
    case mxUINT32_CLASS: return *(const uint32_t *)a->data;
    case mxINT64_CLASS: return (double)*(const int64_t *)a->data;
    case mxUINT64_CLASS: return (double)*(const uint64_t *)a->data;
    default: throw MexError{"fake_mx:scalar", "mxGetScalar of a non-numeric array"};
    }
}
void *mxGetData(const mxArray *a) { return a ? a->data : nullptr; }
double *mxGetPr(const mxArray *a) {
    if (a && a->cls != mxDOUBLE_CLASS) throw MexError{"fake_mx:class", "mxGetPr of a non-double array"};   // (R2018a+ semantics)
    return a ? (double *)a->data : nullptr;
}
size_t mxGetM(const mxArray *a) { return a ? a->m : 0; }
size_t mxGetN(const mxArray *a) { return a ? a->n : 0; }
size_t mxGetNumberOfElements(const mxArray *a) { return a ? a->m * a->n : 0; }
size_t mxGetElementSize(const mxArray *a) { return a ? elem_size(a->cls) : 0; }
mxClassID mxGetClassID(const mxArray *a) { return a ? a->cls : mxUNKNOWN_CLASS; }
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID c, mxComplexity cx) {
    if (cx != mxREAL) throw MexError{"fake_mx:complex", "complex arrays are not modelled"};
    return make(c, m, n);
}
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity cx) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, cx); }
mxArray *mxCreateDoubleScalar(double v) {
    mxArray *a = make(mxDOUBLE_CLASS, 1, 1);
    *(double *)a->data = v;
    return a;
}
mxArray *mxCreateString(const char *s) {
    const size_t k = strlen(s);
    mxArray *a = make(mxCHAR_CLASS, k ? 1 : 0, k);
    for (size_t i = 0; i < k; ++i) ((mxChar *)a->data)[i] = (mxChar)(unsigned char)s[i];
    return a;
}
void mxDestroyArray(mxArray *a) {
    if (!a) return;
    free(a->data);
    delete a;
    --g_live;
}
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw MexError{id ? id : "", buf};
}
void mexLock(void) { ++g_lock; }
void mexUnlock(void) { if (g_lock > 0) --g_lock; }
bool mexIsLocked(void) { return g_lock > 0; }

// ---- driver side (tests/fake_matlab.py, ctypes) -------------------------------------------------------------------------
// a numeric / char array of class `cls`, m x n, filled from `colmajor` (m * n elements, may be NULL: zeros)
mxArray *fm_array(int cls, size_t m, size_t n, const void *colmajor) {
    try {
        mxArray *a = make((mxClassID)cls, m, n);
        if (colmajor && m * n) memcpy(a->data, colmajor, m * n * elem_size((mxClassID)cls));
        return a;
    } catch (...) { return nullptr; }
}
mxArray *fm_string(const char *s) { return mxCreateString(s); }
int fm_class(const mxArray *a) { return (int)mxGetClassID(a); }
size_t fm_m(const mxArray *a) { return mxGetM(a); }
size_t fm_n(const mxArray *a) { return mxGetN(a); }
void *fm_data(const mxArray *a) { return mxGetData(a); }
void fm_free(mxArray *a) { mxDestroyArray(a); }
long fm_live(void) { return g_live.load(); }
int fm_lock_depth(void) { return g_lock; }
// call the gateway; 0 = returned, 1 = mexErrMsgIdAndTxt (id / message copied out), 2 = another C++ exception.
// Like MATLAB, outputs the MEX file had already created when it raised are destroyed.
int fm_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs, char *err_id, char *err_msg, size_t cap) {
    for (int i = 0; i < nlhs; ++i) plhs[i] = nullptr;
    auto put = [&](char *dst, const std::string &s) { if (dst && cap) { strncpy(dst, s.c_str(), cap - 1); dst[cap - 1] = 0; } };
    try {
        // (MATLAB always provides room for one output — `ans` — even when nlhs = 0)
        mxArray *one[1] = {nullptr};
        mexFunction(nlhs, nlhs > 0 ? plhs : one, nrhs, prhs);
        if (nlhs == 0 && one[0]) mxDestroyArray(one[0]);
        return 0;
    } catch (const MexError &e) {
        put(err_id, e.id); put(err_msg, e.msg);
        for (int i = 0; i < nlhs; ++i) if (plhs[i]) { mxDestroyArray(plhs[i]); plhs[i] = nullptr; }
        return 1;
    } catch (const std::exception &e) {
        put(err_id, "fake_mx:exception"); put(err_msg, e.what());
        return 2;
    } catch (...) {
        put(err_id, "fake_mx:exception"); put(err_msg, "unknown C++ exception");
        return 2;
    }
}
}
