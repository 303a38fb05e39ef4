// PolarCode.hpp — header-only C++ host mirror of the reference class (PolarC/PolarCode.h:17-36)
// over the C-ABI of include/polar_amd.h. Same constructor and member names, same argument
// meaning; a translation unit written against the reference's PolarCode.h compiles against this
// header unchanged and runs on the MI355X (link with -lpolar_amd). Errors, which the reference
// turns into std::out_of_range / terminate, surface as std::runtime_error carrying
// polar_last_error().
#ifndef POLAR_AMD_POLARCODE_HPP
#define POLAR_AMD_POLARCODE_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "polar_amd.h"

class PolarCode {
public:
    // PolarCode.h:19-28
    PolarCode(uint8_t num_layers, uint16_t info_length, double epsilon, uint16_t crc_size)
        : _n(num_layers), _info_length(info_length), _crc_size(crc_size) {
        check(polar_create(num_layers, info_length, epsilon, crc_size, &_h));
        _block_length = (uint16_t)(1u << _n);
    }
    // explicit tables (frozen mask, info order, CRC matrix)
    PolarCode(uint8_t num_layers, uint16_t info_length, uint16_t crc_size, const std::vector<uint8_t> &frozen,
              const std::vector<uint16_t> &order, const std::vector<uint8_t> &crc_matrix)
        : _n(num_layers), _info_length(info_length), _crc_size(crc_size) {
        check(polar_create_explicit(num_layers, info_length, crc_size, frozen.data(), order.data(),
                                    crc_size ? crc_matrix.data() : nullptr, &_h));
        _block_length = (uint16_t)(1u << _n);
    }
    ~PolarCode() { polar_destroy(_h); }
    PolarCode(const PolarCode &) = delete;
    PolarCode &operator=(const PolarCode &) = delete;

    // PolarCode.h:30-34
    std::vector<uint8_t> encode(std::vector<uint8_t> info_bits) {
        need(info_bits.size() == _info_length, "encode: info_bits must have info_length entries");
        std::vector<uint8_t> coded(_block_length);
        check(polar_encode(_h, info_bits.data(), coded.data()));
        return coded;
    }
    std::vector<uint8_t> decode_scl_p1(std::vector<double> p1, std::vector<double> p0, uint16_t list_size) {
        need(p1.size() == _block_length && p0.size() == _block_length, "decode_scl_p1: need block_length values");
        std::vector<uint8_t> out(_info_length);
        check(polar_decode_scl_p1(_h, p1.data(), p0.data(), list_size, out.data()));
        return out;
    }
    std::vector<uint8_t> decode_scl_llr(std::vector<double> llr, uint16_t list_size) {
        need(llr.size() == _block_length, "decode_scl_llr: need block_length values");
        std::vector<uint8_t> out(_info_length);
        check(polar_decode_scl_llr(_h, llr.data(), list_size, out.data()));
        return out;
    }
    // PolarM's SC decoder (PolarCode.m:290)
    std::vector<double> decode_sc_p1(std::vector<double> p1) {
        need(p1.size() == _block_length, "decode_sc_p1: need block_length values");
        std::vector<double> out(_info_length);
        check(polar_decode_sc_p1(_h, p1.data(), out.data()));
        return out;
    }
    // batched variant (B codewords, row-major) — the form that fills the GPU
    std::vector<uint8_t> decode_scl_llr_batch(const std::vector<double> &llr, uint16_t list_size) {
        need(llr.size() % _block_length == 0, "decode_scl_llr_batch: size must be a multiple of block_length");
        long B = (long)(llr.size() / _block_length);
        std::vector<uint8_t> out((size_t)B * _info_length);
        check(polar_decode_scl_llr_batch(_h, llr.data(), B, list_size, out.data()));
        return out;
    }
    // single-precision LLRs (B codewords back to back): widened exactly on the device
    std::vector<uint8_t> decode_scl_llr_batch(const std::vector<float> &llr, uint16_t list_size) {
        need(llr.size() % _block_length == 0, "decode_scl_llr_batch: size must be a multiple of block_length");
        const long B = (long)(llr.size() / _block_length);
        std::vector<uint8_t> out((size_t)B * _info_length);
        check(polar_decode_scl_llr_batch_f32(_h, llr.data(), B, list_size, out.data()));
        return out;
    }

    // PolarCode.cpp:658: bler[list_index][ebno_index]; reference constants max_err=100, max_runs=1000
    // PolarCode.h:32-34. `devices` (optional): shard the trials of every round over these GPUs of the node
    // (polar_get_bler_quick_multi: one RCCL all-reduce of the counters per round); `ber` (optional): PolarM's
    // second output (PolarCode.m:781, 848), same layout as the result.
    std::vector<std::vector<double>> get_bler_quick(std::vector<double> ebno_vec, std::vector<uint8_t> list_size,
                                                    long max_runs = 1000, long max_err = 100, uint64_t seed = 1,
                                                    long batch = 0, std::vector<int> devices = {},
                                                    std::vector<std::vector<double>> *ber = nullptr, int constellation = 0) {
        // constellation: 0 = BPSK / Eb/N0 axis (PolarCode.cpp:744-753); POLAR_CONST_ASK{4,8,16}_GRAY (polar_synth.h) = the ASK
        // Gray + BICM sweep of PolarM/main_MC_CC_Comparison.m:44-119 with `ebno_vec` read as the SNR axis in dB
        std::vector<double> flat(ebno_vec.size() * list_size.size()), fber(flat.size());
        check(polar_get_bler_quick_multi_ex(_h, constellation, devices.empty() ? nullptr : devices.data(),
                                            devices.empty() ? 1 : (int)devices.size(), ebno_vec.data(), (int)ebno_vec.size(),
                                            list_size.data(), (int)list_size.size(), max_runs, max_err, seed, batch,   // batch 0: library rounds
                                            flat.data(), fber.data(), nullptr, nullptr, nullptr, nullptr));
        std::vector<std::vector<double>> bler(list_size.size(), std::vector<double>(ebno_vec.size()));
        if (ber) ber->assign(list_size.size(), std::vector<double>(ebno_vec.size()));
        for (size_t l = 0; l < list_size.size(); ++l)
            for (size_t e = 0; e < ebno_vec.size(); ++e) {
                bler[l][e] = flat[l * ebno_vec.size() + e];
                if (ber) (*ber)[l][e] = fber[l * ebno_vec.size() + e];
            }
        return bler;
    }
    polar_code_t *handle() { return _h; }

private:
    static void check(int rc) {
        if (rc < 0) throw std::runtime_error(std::string("polar_amd: ") + polar_last_error());
    }
    static void need(bool ok, const char *msg) {
        if (!ok) throw std::out_of_range(msg);   // the reference throws out_of_range from .at()
    }
    polar_code_t *_h = nullptr;
    uint8_t _n;
    uint16_t _info_length, _block_length = 0, _crc_size;
};

#endif
