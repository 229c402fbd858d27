// sdrxbench -- the counterpart of the reference's `sdrangelbench` (sdrbench/mainbench.cpp:41-391, parserbench.cpp:24-130) on the
// GPU classes of include/sdrx/dsp.hpp: same options, same test types, same result line.
//
//   sdrxbench -t decimateii|decimateinfii|decimatesupii|decimatefi|decimateff|decimateif  -n <samples>  -r <repetitions>  -l <log2>
//             [--device N]
//
// Like the reference it times the decimateK_x(&it, buf, len) CALL, buffers in host memory: here that is PCIe in + kernels +
// PCIe out (the host-pointer C ABI).  A second line gives the same work with the input resident in HBM (sdrx_*_process_dev),
// which is what bench.py reports.  Test data as in the reference: std::mt19937 default seed, int16 uniform in [-2048, 2047],
// float uniform in [-1, 1), the last element left at 0 (mainbench.cpp:76-79, 146-149).
//
//   make -C sdrbench        (hipcc, host code only; links libsdrx.so)
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "sdrx/dsp.hpp"

typedef int16_t qint16;
typedef int32_t qint32;

namespace {

struct Options { std::string test = "decimateii"; int nbSamples = 1048576; int repetition = 1; int log2 = 4; int device = 0; bool hash = false; };

// QCommandLineParser's letters and long names (parserbench.cpp:24-42); invalid values fall back to the defaults with a warning
bool parse(int argc, char** argv, Options& o)
{
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char* s, const char* l) -> const char* {
            if ((a == s || a == l) && i + 1 < argc) return argv[++i];
            const std::string pre = std::string(l) + "=";
            if (a.compare(0, pre.size(), pre) == 0) return argv[i] + pre.size();
            return nullptr;
        };
        if (a == "-h" || a == "--help") {
            printf("Usage: sdrxbench [options]\nSoftware Defined Radio application benchmarks (GPU classes)\n\nOptions:\n"
                   "  -t, --test <test>              Test type.\n  -n, --nb-samples <samples>     Number of sample to deal with.\n"
                   "  -r, --repeat <repetition>      Number of repetitions.\n  -l, --log2-factor <log2>       Log2 factor for rate conversion.\n"
                   "      --device <n>               HIP device.\n      --hash                     print the FNV-1a-64 of the last repetition's output bytes.\n");
            return false;
        }
        if (const char* v = val("-t", "--test")) {
            bool ok = *v != 0; for (const char* p = v; *p; p++) ok = ok && *p >= 'a' && *p <= 'z';
            if (ok) o.test = v; else fprintf(stderr, "ParserBench::parse: test string invalid. Defaulting to %s\n", o.test.c_str());
        } else if (const char* v = val("-n", "--nb-samples")) {
            const long n = strtol(v, nullptr, 10);
            if (n > 1024 && n < 1073741824) o.nbSamples = (int)n; else fprintf(stderr, "ParserBench::parse: number of samples invalid. Defaulting to %d\n", o.nbSamples);
        } else if (const char* v = val("-r", "--repeat")) {
            const long n = strtol(v, nullptr, 10);
            if (n >= 0) o.repetition = (int)n; else fprintf(stderr, "ParserBench::parse: repetition invalid. Defaulting to %d\n", o.repetition);
        } else if (const char* v = val("-l", "--log2-factor")) {
            const long n = strtol(v, nullptr, 10);
            if (n >= 0 && n <= 6) o.log2 = (int)n; else fprintf(stderr, "ParserBench::parse: log2 factor invalid. Defaulting to %d\n", o.log2);
        } else if (a == "--hash") o.hash = true;
        else if (const char* v = val("--device", "--device")) o.device = atoi(v);
        else { fprintf(stderr, "sdrxbench: unknown option %s\n", a.c_str()); return false; }
    }
    return true;
}

void printResults(const char* prefix, const Options& o, double nsecs)
{
    const double ratekSs = ((double)o.nbSamples * o.repetition / nsecs) * 1e6;     // MainBench::printResults (mainbench.cpp:385-391)
    printf("%s: ran test in %.0f ns - sample rate: %g kS/s\n", prefix, nsecs, ratekSs);
}

// FNV-1a-64 over the output bytes: the known-answer check against the CPU oracle on the same (libstdc++-generated) input
void printHash(const Options& o, const void* p, size_t bytes, size_t n)
{
    if (!o.hash) return;
    uint64_t h = 0xcbf29ce484222325ull;
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < bytes; i++) h = (h ^ b[i]) * 0x100000001b3ull;
    printf("hash: test %s log2 %d n %zu fnv1a64 %016llx\n", o.test.c_str(), o.log2, n, (unsigned long long)h);
}

template <typename F> double timed(int reps, F&& f)
{
    double ns = 0;
    for (int i = 0; i < reps; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    }
    return ns;
}

// decimateK_x by log2 factor, as MainBench::decimateII / InfII / SupII / FI / FF / IF switch on it (mainbench.cpp:224-383)
template <typename Dec, typename It, typename T> void by_log2(Dec& d, int fc, int log2, It* it, const T* buf, int len)
{
#define K(L, NAME) case L: if (fc == 0) d.NAME##_inf(it, buf, len); else if (fc == 1) d.NAME##_sup(it, buf, len); else d.NAME##_cen(it, buf, len); break;
    switch (log2) {
    case 0: d.decimate1(it, buf, len); break;
    K(1, decimate2) K(2, decimate4) K(3, decimate8) K(4, decimate16) K(5, decimate32) K(6, decimate64)
    default: break;
    }
#undef K
}

// the same work with the input resident in HBM: one warm-up call, then `reps` timed calls behind a sync
template <typename Create, typename Run, typename Sync>
double resident(const void* host, size_t bytes, size_t out_bytes, int reps, Create create, Run run, Sync sync)
{
    void* d_in = nullptr; void* d_out = nullptr;
    if (hipMalloc(&d_in, bytes) != hipSuccess || hipMalloc(&d_out, out_bytes + 64) != hipSuccess) return -1;
    hipMemcpy(d_in, host, bytes, hipMemcpyHostToDevice);
    void* h = create();
    if (!h) return -1;
    run(h, d_in, d_out); sync(h);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) run(h, d_in, d_out);
    sync(h);
    const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    hipFree(d_in); hipFree(d_out);
    return ns;
}

} // namespace

int main(int argc, char** argv)
{
    Options o;
    if (!parse(argc, argv, o)) return 1;
    if (sdrx_device_count() <= o.device) { fprintf(stderr, "sdrxbench: no HIP device %d (this bench has no CPU path)\n", o.device); return 2; }
    std::mt19937 gen;                                                   // default-seeded, as MainBench's m_generator
    std::uniform_real_distribution<float> dist_f(-1.0, 1.0);
    std::uniform_int_distribution<qint16> dist_s16(-2048, 2047);
    const size_t n2 = (size_t)o.nbSamples * 2;
    const int fc = o.test == "decimateinfii" ? 0 : o.test == "decimatesupii" ? 1 : 2;
    const int reps = o.repetition;
    const size_t n_out = (size_t)o.nbSamples >> o.log2;

    if (o.test == "decimatefi" || o.test == "decimateff") {
        std::vector<float> buf(n2, 0.0f);
        std::generate(buf.begin(), buf.end() - 1, std::bind(dist_f, gen));
        const bool ff = o.test == "decimateff";
        double ns;
        if (ff) {
            sdrx::DecimatorsFF dec(o.device); sdrx::FSampleVector out(n_out + 1);
            size_t k = 0;
            ns = timed(reps, [&] { sdrx::FSampleVector::iterator it = out.begin(); by_log2(dec, 2, o.log2, &it, buf.data(), (int)n2); k = (size_t)(it - out.begin()); });
            printHash(o, out.data(), k * sizeof(out[0]), k);
        } else {
            sdrx::DecimatorsFI dec(o.device); sdrx::SampleVector out(n_out + 1);
            size_t k = 0;
            ns = timed(reps, [&] { sdrx::SampleVector::iterator it = out.begin(); by_log2(dec, 2, o.log2, &it, buf.data(), (int)n2); k = (size_t)(it - out.begin()); });
            printHash(o, out.data(), k * sizeof(out[0]), k);
        }
        printResults(ff ? "MainBench::testDecimateFF" : "MainBench::testDecimateFI", o, ns);
        const double r = resident(buf.data(), n2 * 4, n_out * 8, reps,
            [&]() -> void* { sdrx_fdecim_t* h = nullptr; sdrx_fdecim_create(&h, o.device, o.log2, SDRX_FC_CEN, SDRX_FD_IN_F32, ff ? SDRX_FD_OUT_F32 : SDRX_FD_OUT_I16, 16); return h; },
            [&](void* h, void* in, void* out) { int64_t k; sdrx_fdecim_process_dev(static_cast<sdrx_fdecim_t*>(h), in, (int64_t)n2, out, &k); },
            [&](void* h) { sdrx_fdecim_sync(static_cast<sdrx_fdecim_t*>(h)); });
        if (r > 0) printResults("  input resident in HBM", o, r);
    } else if (o.test == "decimateif") {
        std::vector<qint16> buf(n2, 0);
        std::generate(buf.begin(), buf.end() - 1, std::bind(dist_s16, gen));
        sdrx::DecimatorsIF<qint16, 12> dec(o.device); sdrx::FSampleVector out(n_out + 1);
        size_t k = 0;
        const double ns = timed(reps, [&] { sdrx::FSampleVector::iterator it = out.begin(); by_log2(dec, 2, o.log2, &it, buf.data(), (int)n2); k = (size_t)(it - out.begin()); });
        printHash(o, out.data(), k * sizeof(out[0]), k);
        printResults("MainBench::testDecimateIF", o, ns);
        const double r = resident(buf.data(), n2 * 2, n_out * 8, reps,
            [&]() -> void* { sdrx_fdecim_t* h = nullptr; sdrx_fdecim_create(&h, o.device, o.log2, SDRX_FC_CEN, SDRX_FD_IN_I16, SDRX_FD_OUT_F32, 12); return h; },
            [&](void* h, void* in, void* out) { int64_t k; sdrx_fdecim_process_dev(static_cast<sdrx_fdecim_t*>(h), in, (int64_t)n2, out, &k); },
            [&](void* h) { sdrx_fdecim_sync(static_cast<sdrx_fdecim_t*>(h)); });
        if (r > 0) printResults("  input resident in HBM", o, r);
    } else {
        std::vector<qint16> buf(n2, 0);
        std::generate(buf.begin(), buf.end() - 1, std::bind(dist_s16, gen));
        sdrx::Decimators<qint32, qint16, 16, 12> dec(o.device); sdrx::SampleVector out(n_out + 1);
        size_t k = 0;
        const double ns = timed(reps, [&] { sdrx::SampleVector::iterator it = out.begin(); by_log2(dec, fc, o.log2, &it, buf.data(), (int)n2); k = (size_t)(it - out.begin()); });
        printHash(o, out.data(), k * sizeof(out[0]), k);
        printResults("MainBench::testDecimateII", o, ns);
        const double r = resident(buf.data(), n2 * 2, n_out * 4, reps,
            [&]() -> void* { sdrx_decim_t* h = nullptr; sdrx_decim_create(&h, o.device, o.log2, fc, 12); return h; },
            [&](void* h, void* in, void* out) { int64_t k; sdrx_decim_process_dev(static_cast<sdrx_decim_t*>(h), static_cast<const int16_t*>(in), (int64_t)n2, static_cast<int16_t*>(out), &k); },
            [&](void* h) { sdrx_decim_sync(static_cast<sdrx_decim_t*>(h)); });
        if (r > 0) printResults("  input resident in HBM", o, r);
    }
    return 0;
}
