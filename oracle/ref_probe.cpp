// oracle/ref_probe.cpp — TEST INFRASTRUCTURE ONLY (our code, not the reference's).
//
// A tiny stdin/stdout driver that #includes the reference's headers *in place* (from
// /root/reference/src via -I, see oracle/Makefile target `ref`) so that golden vectors for the
// primitives that the pybind module does not expose can be produced by the reference itself:
//   jenkins   : emphf::jenkins64_hasher::operator()(byte_range)   src/emphf/base_hash.hpp:38-91
//   lookup    : emphf::mphf<jenkins64_hasher>::lookup             src/emphf/mphf.hpp:79-89
//   enc23/13  : get_dna23_bitset / get_dna13_bitset               src/kmers.cpp:12-55
//   rev23/13  : reverseDNA(u64) / reverseDNA(u32)                 src/kmers.cpp:376-388
// Used only by tests/golden/make_golden.py inside the build container.
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

#include "emphf/common.hpp"
#include "emphf/base_hash.hpp"
#include "emphf/mphf.hpp"
#include "kmers.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: ref_probe jenkins|lookup <pf>|enc23|enc13|rev23|rev13  < stdin\n");
        return 2;
    }
    std::string mode = argv[1];
    emphf::stl_string_adaptor adaptor;
    if (mode == "jenkins") {  // lines: "<seed-hex> <string>"
        std::string seed_s, s;
        while (std::cin >> seed_s >> s) {
            uint64_t seed = std::stoull(seed_s, nullptr, 16);
            emphf::jenkins64_hasher h(seed);
            auto t = h(adaptor(s));
            std::printf("%016llx %016llx %016llx\n", (unsigned long long)std::get<0>(t),
                        (unsigned long long)std::get<1>(t), (unsigned long long)std::get<2>(t));
        }
    } else if (mode == "lookup") {  // lines: "<string>"
        if (argc < 3) return 2;
        emphf::mphf<emphf::jenkins64_hasher> f;
        std::ifstream is(argv[2], std::ios::binary);
        if (!is) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        f.load(is);
        std::string s;
        while (std::cin >> s) std::printf("%llu\n", (unsigned long long)f.lookup(s, adaptor));
    } else if (mode == "enc23") {
        std::string s;
        while (std::cin >> s) std::printf("%llu\n", (unsigned long long)get_dna23_bitset(s));
    } else if (mode == "enc13") {
        std::string s;
        while (std::cin >> s) std::printf("%u\n", (unsigned)get_dna13_bitset(s));
    } else if (mode == "rev23") {
        unsigned long long v;
        while (std::cin >> v) std::printf("%llu\n", (unsigned long long)reverseDNA((uint64_t)v));
    } else if (mode == "rev13") {
        unsigned long long v;
        while (std::cin >> v) std::printf("%u\n", (unsigned)reverseDNA((uint32_t)v));
    } else {
        return 2;
    }
    return 0;
}
