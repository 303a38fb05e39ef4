// minimal mex.h stand-in for a COMPILE-ONLY syntax check of polar_mex.cpp (tests/test_abi.py); not a MATLAB binding
#pragma once
#include <cstddef>
#include <cstdint>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxUINT8_CLASS, mxUINT16_CLASS, mxUINT64_CLASS, mxDOUBLE_CLASS, mxINT32_CLASS, mxSINGLE_CLASS } mxClassID;
typedef enum { mxREAL } mxComplexity;
extern "C" {
bool mxIsChar(const mxArray *); int mxGetString(const mxArray *, char *, mwSize); double mxGetScalar(const mxArray *);
void *mxGetData(const mxArray *); double *mxGetPr(const mxArray *); size_t mxGetM(const mxArray *); size_t mxGetN(const mxArray *);
size_t mxGetNumberOfElements(const mxArray *); mxClassID mxGetClassID(const mxArray *); mxArray *mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxArray *mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity); mxArray *mxCreateDoubleScalar(double);
void mexErrMsgIdAndTxt(const char *, const char *, ...); void mexLock(void); void mexUnlock(void);
}
