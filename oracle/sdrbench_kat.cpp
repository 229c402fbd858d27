// TEST INFRASTRUCTURE ONLY.  Known answers for sdrbench/sdrxbench: the CPU oracle on sdrangelbench's own test data
// (std::mt19937 default seed, libstdc++ distributions, last element 0: sdrbench/mainbench.cpp:76-79, 146-149), printed in the
// format of `sdrxbench --hash`.  usage: sdrbench_kat <test> <log2> [nb_samples [repetitions]]
// (the repetitions run on ONE decimator object, like the reference's: the filters carry their state into the next one)
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>
extern "C" {
#include "sdro.h"
}

int main(int argc, char** argv)
{
    if (argc < 3) return 1;
    const std::string test = argv[1];
    const int log2 = atoi(argv[2]);
    const int n = argc > 3 ? atoi(argv[3]) : 1048576;
    const int reps = argc > 4 ? atoi(argv[4]) : 1;
    std::mt19937 gen;
    std::uniform_real_distribution<float> dist_f(-1.0, 1.0);
    std::uniform_int_distribution<int16_t> dist_s16(-2048, 2047);
    const size_t n2 = (size_t)n * 2;
    std::vector<unsigned char> out;
    size_t k = 0;
    if (test == "decimatefi" || test == "decimateff") {
        std::vector<float> buf(n2, 0.0f);
        std::generate(buf.begin(), buf.end() - 1, std::bind(dist_f, gen));
        const bool ff = test == "decimateff";
        sdro_fdecim* d = sdro_fdecim_new(log2, SDRO_FC_CEN, 0, ff ? 1 : 0, 16);
        out.resize(n2 * 4 + 64);
        for (int r = 0; r < reps; r++) k = (size_t)sdro_fdecim_process(d, buf.data(), (int32_t)n2, out.data());
        out.resize(k * (ff ? 8 : 4));
    } else if (test == "decimateif") {
        std::vector<int16_t> buf(n2, 0);
        std::generate(buf.begin(), buf.end() - 1, std::bind(dist_s16, gen));
        sdro_fdecim* d = sdro_fdecim_new(log2, SDRO_FC_CEN, 1, 1, 12);
        out.resize(n2 * 4 + 64);
        for (int r = 0; r < reps; r++) k = (size_t)sdro_fdecim_process(d, buf.data(), (int32_t)n2, out.data());
        out.resize(k * 8);
    } else {
        std::vector<int16_t> buf(n2, 0);
        std::generate(buf.begin(), buf.end() - 1, std::bind(dist_s16, gen));
        const int fc = test == "decimateinfii" ? SDRO_FC_INF : test == "decimatesupii" ? SDRO_FC_SUP : SDRO_FC_CEN;
        sdro_decim* d = sdro_decim_new(log2, fc, 12);
        out.resize(n2 * 2 + 64);
        for (int r = 0; r < reps; r++) k = (size_t)sdro_decim_process(d, buf.data(), (int32_t)n2, reinterpret_cast<int16_t*>(out.data()));
        out.resize(k * 4);
    }
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char b : out) h = (h ^ b) * 0x100000001b3ull;
    printf("hash: test %s log2 %d n %zu fnv1a64 %016llx\n", test.c_str(), log2, k, (unsigned long long)h);
    return 0;
}
