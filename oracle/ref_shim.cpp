// oracle/ref_shim.cpp — TEST INFRASTRUCTURE, not product code.
//
// Thin extern "C" window onto the UNMODIFIED reference implementation
// (/root/reference/PolarC/PolarCode.{h,cpp}), compiled where it lies by
// oracle/Makefile into oracle/_ref/libpolarc_ref.so.  Nothing from the
// reference is copied: this file only includes its header by path and
// forwards calls.  Used to (1) pin oracle/polar_oracle.c, (2) generate the
// golden vectors under tests/golden/, (3) serve as the "reference" CPU
// baseline in bench.py.  The product (polar_amd/) never links or loads it.
//
// `#define private public` is the SURVEY §8c trick to read the private
// tables (frozen mask, channel order, CRC matrix) without touching the
// reference source.
#include <sstream>
#include <fstream>
#include <iostream>
#include <random>
#include <algorithm>
#include <chrono>
#include <stack>
#include <vector>
#define private public
#include "PolarCode.h"
#undef private
#include <cstring>
#include <cstdlib>

extern "C" {

// PolarCode::PolarCode (PolarCode.h:19-28). NOTE: consumes crc*K values of
// the process-global glibc rand() stream (PolarCode.cpp:51-56).
void *ref_create(int n, int K, double eps, int crc) {
    return new PolarCode((uint8_t)n, (uint16_t)K, eps, (uint16_t)crc);
}
void ref_destroy(void *h) { delete (PolarCode *)h; }
void ref_srand(unsigned s) { srand(s); }

int ref_block_length(void *h) { return ((PolarCode *)h)->_block_length; }

// tables (PolarCode.h:45-48)
void ref_get_frozen(void *h, uint8_t *out) {
    PolarCode *p = (PolarCode *)h;
    memcpy(out, p->_frozen_bits.data(), p->_block_length);
}
void ref_get_order(void *h, uint16_t *out) {
    PolarCode *p = (PolarCode *)h;
    memcpy(out, p->_channel_order_descending.data(), 2 * (size_t)p->_block_length);
}
void ref_get_bitrev(void *h, uint16_t *out) {
    PolarCode *p = (PolarCode *)h;
    memcpy(out, p->_bit_rev_order.data(), 2 * (size_t)p->_block_length);
}
void ref_get_crc_matrix(void *h, uint8_t *out) {  // row-major crc x K
    PolarCode *p = (PolarCode *)h;
    for (unsigned i = 0; i < p->_crc_size; ++i)
        memcpy(out + (size_t)i * p->_info_length, p->_crc_matrix[i].data(), p->_info_length);
}
// overwrite the CRC matrix (lets a fixture pin one matrix independent of rand() state)
void ref_set_crc_matrix(void *h, const uint8_t *in) {
    PolarCode *p = (PolarCode *)h;
    for (unsigned i = 0; i < p->_crc_size; ++i)
        memcpy(p->_crc_matrix[i].data(), in + (size_t)i * p->_info_length, p->_info_length);
}
// overwrite frozen mask + order (config 5: Monte-Carlo constructed code, PolarCode.m:111-135)
void ref_set_tables(void *h, const uint8_t *frozen, const uint16_t *order) {
    PolarCode *p = (PolarCode *)h;
    memcpy(p->_frozen_bits.data(), frozen, p->_block_length);
    memcpy(p->_channel_order_descending.data(), order, 2 * (size_t)p->_block_length);
}

// PolarCode::encode (PolarCode.cpp:60-91)
void ref_encode(void *h, const uint8_t *info, uint8_t *coded) {
    PolarCode *p = (PolarCode *)h;
    std::vector<uint8_t> v(info, info + p->_info_length);
    std::vector<uint8_t> c = p->encode(v);
    memcpy(coded, c.data(), c.size());
}
// PolarCode::decode_scl_llr (PolarCode.cpp:130-148)
void ref_decode_scl_llr(void *h, const double *llr, int L, uint8_t *out) {
    PolarCode *p = (PolarCode *)h;
    std::vector<double> v(llr, llr + p->_block_length);
    std::vector<uint8_t> d = p->decode_scl_llr(v, (uint16_t)L);
    memcpy(out, d.data(), d.size());
}
void ref_decode_scl_llr_batch(void *h, const double *llr, long B, int L, uint8_t *out) {
    PolarCode *p = (PolarCode *)h;
    for (long b = 0; b < B; ++b)
        ref_decode_scl_llr(h, llr + b * (long)p->_block_length, L, out + b * (long)p->_info_length);
}
// PolarCode::decode_scl_p1 (PolarCode.cpp:110-128)
void ref_decode_scl_p1(void *h, const double *p1, const double *p0, int L, uint8_t *out) {
    PolarCode *p = (PolarCode *)h;
    std::vector<double> a(p1, p1 + p->_block_length), b(p0, p0 + p->_block_length);
    std::vector<uint8_t> d = p->decode_scl_p1(a, b, (uint16_t)L);
    memcpy(out, d.data(), d.size());
}
// PolarCode::get_bler_quick (PolarCode.cpp:658-785); bler_out is [n_L][n_e] row-major.
// The reference prints progress on std::cout; silence it for the caller.
void ref_get_bler_quick(void *h, const double *ebno, int n_e, const uint8_t *L, int n_L, double *bler_out) {
    PolarCode *p = (PolarCode *)h;
    std::vector<double> e(ebno, ebno + n_e);
    std::vector<uint8_t> l(L, L + n_L);
    std::streambuf *old = std::cout.rdbuf();
    std::ostringstream sink;
    std::cout.rdbuf(sink.rdbuf());
    std::vector<std::vector<double>> r = p->get_bler_quick(e, l);
    std::cout.rdbuf(old);
    for (int i = 0; i < n_L; ++i)
        for (int j = 0; j < n_e; ++j) bler_out[i * n_e + j] = r[i][j];
}

}  // extern "C"
