// tests/mex_runtime/mex.h — a small WORKING stand-in for MATLAB's MEX / mx C API (the subset polar_mex.cpp uses), so that the
// gateway polar_amd/matlab/polar_mex.cpp can be compiled, LINKED and RUN without MATLAB (which is not in the image): tests only.
// Semantics follow MathWorks' documented behaviour for each call: column-major storage, class ids with MATLAB's own enum
// values, zero-initialised numeric arrays, mxGetScalar converting the first element to double, char arrays of 16-bit mxChar,
// mexErrMsgIdAndTxt never returning (here: a C++ exception the test driver catches — MATLAB longjmps out of the MEX file).
// Implementation: tests/mex_runtime/mex_runtime.cpp; Python driver: tests/fake_matlab.py.
#pragma once
#include <cstddef>
#include <cstdint>

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef uint16_t mxChar;
typedef enum {             // matrix.h values
    mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS, mxSINGLE_CLASS,
    mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS
} mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;

extern "C" {
bool mxIsChar(const mxArray *);
int mxGetString(const mxArray *, char *, mwSize);          // 0 on success, 1 when the buffer is too small / not a char array
double mxGetScalar(const mxArray *);
void *mxGetData(const mxArray *);
double *mxGetPr(const mxArray *);
size_t mxGetM(const mxArray *);
size_t mxGetN(const mxArray *);
size_t mxGetNumberOfElements(const mxArray *);
size_t mxGetElementSize(const mxArray *);
mxClassID mxGetClassID(const mxArray *);
mxArray *mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);     // zero-initialised, like MATLAB's
mxArray *mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray *mxCreateDoubleScalar(double);
mxArray *mxCreateString(const char *);
void mxDestroyArray(mxArray *);
void mexErrMsgIdAndTxt(const char *, const char *, ...) __attribute__((noreturn));
void mexLock(void);
void mexUnlock(void);
bool mexIsLocked(void);
// the gateway (polar_amd/matlab/polar_mex.cpp)
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
}
