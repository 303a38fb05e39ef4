// main.cpp — CLI counterpart of the reference driver (PolarC/main.cpp:6-43): same code
// (n=11, K=1024, eps=0.32), same Eb/N0 grid and list sizes, same table layout on stdout.
// Options: --crc C --runs R --max-err E --seed S --batch B --emin x --emax x --estep x --L "1,2,4"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <vector>

#include "PolarCode.hpp"

int main(int argc, char *argv[]) {
    uint8_t n = 11;
    uint16_t crc_size = 0;
    long runs = 1000, max_err = 100, batch = 0;
    uint64_t seed = 1;
    double emin = 1.00, emax = 2.01, estep = 0.25, eps = 0.32;
    std::vector<uint8_t> list_size_vec = {1, 2, 4, 8, 32};
    int K = -1;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string a = argv[i];
        const char *v = argv[i + 1];
        if (a == "--n") n = (uint8_t)atoi(v);
        else if (a == "--K") K = atoi(v);
        else if (a == "--crc") crc_size = (uint16_t)atoi(v);
        else if (a == "--runs") runs = atol(v);
        else if (a == "--max-err") max_err = atol(v);
        else if (a == "--batch") batch = atol(v);
        else if (a == "--seed") seed = strtoull(v, nullptr, 10);
        else if (a == "--emin") emin = atof(v);
        else if (a == "--emax") emax = atof(v);
        else if (a == "--estep") estep = atof(v);
        else if (a == "--L") {
            list_size_vec.clear();
            std::stringstream ss(v);
            std::string tok;
            while (std::getline(ss, tok, ',')) list_size_vec.push_back((uint8_t)atoi(tok.c_str()));
        } else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    uint16_t info_length = (uint16_t)(K > 0 ? K : (1 << (n - 1)));
    try {
        PolarCode polar_code(n, info_length, eps, crc_size);
        std::vector<double> ebno_vec;
        for (double e = emin; e <= emax; e += estep) ebno_vec.push_back(e);
        std::vector<std::vector<double>> bler =
            polar_code.get_bler_quick(ebno_vec, list_size_vec, runs, max_err, seed, batch);
        for (unsigned i_ebno = 0; i_ebno < ebno_vec.size(); ++i_ebno) {
            std::cout << std::fixed << std::setprecision(3) << ebno_vec.at(i_ebno) << "\t \t";
            for (unsigned i_list = 0; i_list < list_size_vec.size(); ++i_list)
                std::cout << std::fixed << std::setprecision(6) << bler.at(i_list).at(i_ebno) << "\t";
            std::cout << std::endl;
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
