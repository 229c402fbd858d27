// C++ mirror of the reference's sdrbase/dsp classes for the RX hot path, header-only, over the
// C ABI of libsdrx.so (include/sdrx.h).  Same names, argument meaning and (absence of) error
// reporting as the reference so that a device thread / channel plugin compiles against either:
//
//   reference                                              here
//   Sample, SampleVector          (dsp/dsptypes.h:44-97)   sdrx::Sample, sdrx::SampleVector
//   Decimators<qint32,qint16,16,B> (dsp/decimators.h:279)  sdrx::Decimators<int32_t,int16_t,16,B>
//   DownChannelizer               (dsp/downchannelizer.h)  sdrx::DownChannelizerBank (N channels, one stream)
//   SampleSinkFifo                (dsp/samplesinkfifo.h)   sdrx::SampleSinkFifo
//   DecimatorsFI / FF / IF<T,B>   (dsp/decimatorsf*.h, decimatorsif.h)  sdrx::DecimatorsFI / DecimatorsFF / DecimatorsIF<T,B>
//
// The reference keeps ONE set of six stage states per Decimators object, shared by all decimateK_* methods.  Here every
// (K, fcPos) pair is a handle (created on first use) and the object carries a sdrx_decim_stages_t: when a call names another
// variant than the previous one, the old handle's filters are saved into it and the new handle continues from them
// (sdrx_decim_save_stages / sdrx_decim_load_stages), so a change of K or fcPos at run time gives the reference object's
// samples, leftovers of the other cascade included.  DecimatorsFI / FF / IF below do the same with sdrx_fdecim_*_stages.
// Visible difference: DSP calls still return void -- a failing GPU call is logged to stderr and the output iterator does
// not advance (the reference has no error path at all on these calls).
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../sdrx.h"

namespace sdrx {

// Inside the reference tree the classes below must take the reference's own SampleVector iterators: define
// SDRX_HOST_SAMPLE to that type (`#define SDRX_HOST_SAMPLE ::Sample` after including dsp/dsptypes.h) and
// sdrx::Sample becomes an alias of it -- it has to be the packed {int16 re, int16 im} of dsptypes.h:44-65.
#ifdef SDRX_HOST_SAMPLE
typedef SDRX_HOST_SAMPLE Sample;
#else
#pragma pack(push, 1)
struct Sample {                                    // dsp/dsptypes.h:44-65
    Sample() : m_real(0), m_imag(0) {}
    Sample(int16_t real, int16_t imag = 0) : m_real(real), m_imag(imag) {}
    int16_t real() const { return m_real; }
    int16_t imag() const { return m_imag; }
    void setReal(int16_t v) { m_real = v; }
    void setImag(int16_t v) { m_imag = v; }
    int16_t m_real, m_imag;
};
#pragma pack(pop)
#endif
static_assert(sizeof(Sample) == 4, "Sample must be a packed {int16 re, int16 im}");
typedef std::vector<Sample> SampleVector;

template<typename StorageType, typename T, unsigned SdrBits, unsigned InputBits>
class Decimators {
    static_assert(sizeof(T) == 2 && SdrBits == 16, "this build covers Decimators<qint32,qint16,16,{8,12,16}>");
public:
    explicit Decimators(int device = 0) : m_device(device), m_stages(nullptr), m_last(nullptr) { for (auto& row : m_h) for (auto& h : row) h = nullptr; }
    ~Decimators() { for (auto& row : m_h) for (auto& h : row) if (h) sdrx_decim_destroy(h); if (m_stages) sdrx_decim_stages_destroy(m_stages); }
    Decimators(const Decimators&) = delete;
    Decimators& operator=(const Decimators&) = delete;

    void decimate1(SampleVector::iterator* it, const T* buf, int32_t len) { run(0, SDRX_FC_CEN, it, buf, len); }
#define SDRX_DECIM(K, L)                                                                                          \
    void decimate##K##_inf(SampleVector::iterator* it, const T* buf, int32_t len) { run(L, SDRX_FC_INF, it, buf, len); } \
    void decimate##K##_sup(SampleVector::iterator* it, const T* buf, int32_t len) { run(L, SDRX_FC_SUP, it, buf, len); } \
    void decimate##K##_cen(SampleVector::iterator* it, const T* buf, int32_t len) { run(L, SDRX_FC_CEN, it, buf, len); }
    SDRX_DECIM(2, 1) SDRX_DECIM(4, 2) SDRX_DECIM(8, 3) SDRX_DECIM(16, 4) SDRX_DECIM(32, 5) SDRX_DECIM(64, 6)
#undef SDRX_DECIM

private:
    void run(int log2, int fcpos, SampleVector::iterator* it, const T* buf, int32_t len)
    {
        sdrx_decim_t*& h = m_h[log2][fcpos];
        if (!h && sdrx_decim_create(&h, m_device, log2, fcpos, (int)InputBits) != SDRX_OK) {
            std::fprintf(stderr, "sdrx::Decimators: %s\n", sdrx_last_error());
            h = nullptr; return;
        }
        if (log2 > 0 && h != m_last) {                     // another cascade on the same six filters (decimators.h:326-333)
            if (!m_stages && sdrx_decim_stages_create(&m_stages, m_device) != SDRX_OK) { std::fprintf(stderr, "sdrx::Decimators: %s\n", sdrx_last_error()); return; }
            if ((m_last && sdrx_decim_save_stages(m_last, m_stages) != SDRX_OK) || sdrx_decim_load_stages(h, m_stages) != SDRX_OK) {
                std::fprintf(stderr, "sdrx::Decimators: %s\n", sdrx_last_error()); return;
            }
            m_last = h;
        }
        int32_t n = 0;
        // Sample is a packed {int16,int16}: the vector's storage is the output buffer
        if (sdrx_decim_process(h, reinterpret_cast<const int16_t*>(buf), len,
                               reinterpret_cast<int16_t*>(&**it), &n) != SDRX_OK) {
            std::fprintf(stderr, "sdrx::Decimators: %s\n", sdrx_last_error());
            return;
        }
        *it += n;
    }
    int m_device;
    sdrx_decim_t* m_h[7][3];
    sdrx_decim_stages_t* m_stages;
    sdrx_decim_t* m_last;
};

// DecimatorsU<qint32, quint8, 16, 8, Shift> (dsp/decimatorsu.h:175-216), the RTL-SDR thread's member
// (plugins/samplesource/rtlsdr/rtlsdrthread.h:55)
template<typename StorageType, typename T, unsigned SdrBits, unsigned InputBits, int Shift>
class DecimatorsU {
    static_assert(sizeof(T) == 1 && SdrBits == 16 && InputBits == 8, "this build covers DecimatorsU<qint32,quint8,16,8,Shift>");
public:
    explicit DecimatorsU(int device = 0) : m_device(device), m_stages(nullptr), m_last(nullptr) { for (auto& row : m_h) for (auto& h : row) h = nullptr; }
    ~DecimatorsU() { for (auto& row : m_h) for (auto& h : row) if (h) sdrx_decim_destroy(h); if (m_stages) sdrx_decim_stages_destroy(m_stages); }
    DecimatorsU(const DecimatorsU&) = delete;
    DecimatorsU& operator=(const DecimatorsU&) = delete;
    void decimate1(SampleVector::iterator* it, const T* buf, int32_t len) { run(0, SDRX_FC_CEN, it, buf, len); }
#define SDRX_DECIMU(K, L)                                                                                         \
    void decimate##K##_inf(SampleVector::iterator* it, const T* buf, int32_t len) { run(L, SDRX_FC_INF, it, buf, len); } \
    void decimate##K##_sup(SampleVector::iterator* it, const T* buf, int32_t len) { run(L, SDRX_FC_SUP, it, buf, len); } \
    void decimate##K##_cen(SampleVector::iterator* it, const T* buf, int32_t len) { run(L, SDRX_FC_CEN, it, buf, len); }
    SDRX_DECIMU(2, 1) SDRX_DECIMU(4, 2) SDRX_DECIMU(8, 3) SDRX_DECIMU(16, 4) SDRX_DECIMU(32, 5) SDRX_DECIMU(64, 6)
#undef SDRX_DECIMU
private:
    void run(int log2, int fcpos, SampleVector::iterator* it, const T* buf, int32_t len)
    {
        sdrx_decim_t*& h = m_h[log2][fcpos];
        if (!h && sdrx_decim_create_u8(&h, m_device, log2, fcpos, Shift) != SDRX_OK) {
            std::fprintf(stderr, "sdrx::DecimatorsU: %s\n", sdrx_last_error()); h = nullptr; return;
        }
        if (log2 > 0 && h != m_last) {
            if (!m_stages && sdrx_decim_stages_create(&m_stages, m_device) != SDRX_OK) { std::fprintf(stderr, "sdrx::DecimatorsU: %s\n", sdrx_last_error()); return; }
            if ((m_last && sdrx_decim_save_stages(m_last, m_stages) != SDRX_OK) || sdrx_decim_load_stages(h, m_stages) != SDRX_OK) {
                std::fprintf(stderr, "sdrx::DecimatorsU: %s\n", sdrx_last_error()); return;
            }
            m_last = h;
        }
        int32_t n = 0;
        if (sdrx_decim_process_u8(h, reinterpret_cast<const uint8_t*>(buf), len, reinterpret_cast<int16_t*>(&**it), &n) != SDRX_OK) {
            std::fprintf(stderr, "sdrx::DecimatorsU: %s\n", sdrx_last_error()); return;
        }
        *it += n;
    }
    int m_device;
    sdrx_decim_t* m_h[7][3];
    sdrx_decim_stages_t* m_stages;
    sdrx_decim_t* m_last;
};

// ---- float half-band decimators (dsp/decimatorsfi.h, decimatorsff.h, decimatorsif.h) over sdrx_fdecim_*.
// FSample = the reference's {Real re; Real im} (dsptypes.h:67-87); define SDRX_HOST_FSAMPLE like SDRX_HOST_SAMPLE.
#ifdef SDRX_HOST_FSAMPLE
typedef SDRX_HOST_FSAMPLE FSample;
#else
struct FSample {
    FSample() : m_real(0), m_imag(0) {}
    FSample(float real, float imag = 0) : m_real(real), m_imag(imag) {}
    float real() const { return m_real; }
    float imag() const { return m_imag; }
    void setReal(float v) { m_real = v; }
    void setImag(float v) { m_imag = v; }
    float m_real, m_imag;
};
#endif
static_assert(sizeof(FSample) == 8, "FSample must be {float re, float im}");
typedef std::vector<FSample> FSampleVector;

// shared body: one handle per (K, fcPos) method, created on first use, all on one shared filter set (as Decimators above)
template<typename OutVec, typename InT, int IN_KIND, int OUT_KIND, int BITS>
class FloatDecimatorsBase {
public:
    explicit FloatDecimatorsBase(int device = 0) : m_device(device), m_stages(nullptr), m_last(nullptr) { for (auto& row : m_h) for (auto& h : row) h = nullptr; }
    ~FloatDecimatorsBase() { for (auto& row : m_h) for (auto& h : row) if (h) sdrx_fdecim_destroy(h); if (m_stages) sdrx_fdecim_stages_destroy(m_stages); }
    FloatDecimatorsBase(const FloatDecimatorsBase&) = delete;
    FloatDecimatorsBase& operator=(const FloatDecimatorsBase&) = delete;
    void decimate1(typename OutVec::iterator* it, const InT* buf, int32_t nbIAndQ) { run(0, SDRX_FC_CEN, it, buf, nbIAndQ); }
#define SDRX_FDECIM(K, L)                                                                                                     \
    void decimate##K##_inf(typename OutVec::iterator* it, const InT* buf, int32_t nbIAndQ) { run(L, SDRX_FC_INF, it, buf, nbIAndQ); } \
    void decimate##K##_sup(typename OutVec::iterator* it, const InT* buf, int32_t nbIAndQ) { run(L, SDRX_FC_SUP, it, buf, nbIAndQ); } \
    void decimate##K##_cen(typename OutVec::iterator* it, const InT* buf, int32_t nbIAndQ) { run(L, SDRX_FC_CEN, it, buf, nbIAndQ); }
    SDRX_FDECIM(2, 1) SDRX_FDECIM(4, 2) SDRX_FDECIM(8, 3) SDRX_FDECIM(16, 4) SDRX_FDECIM(32, 5) SDRX_FDECIM(64, 6)
#undef SDRX_FDECIM
private:
    void run(int log2, int fcpos, typename OutVec::iterator* it, const InT* buf, int32_t n)
    {
        sdrx_fdecim_t*& h = m_h[log2][fcpos];
        if (!h && sdrx_fdecim_create(&h, m_device, log2, fcpos, IN_KIND, OUT_KIND, BITS) != SDRX_OK) {
            std::fprintf(stderr, "sdrx float decimators: %s\n", sdrx_last_error()); h = nullptr; return;
        }
        if (h != m_last) {                                 // another cascade on the object's six filters (decimatorsfi.h: m_decimator2 .. 64)
            if (!m_stages && sdrx_fdecim_stages_create(&m_stages, m_device) != SDRX_OK) { std::fprintf(stderr, "sdrx float decimators: %s\n", sdrx_last_error()); return; }
            if ((m_last && sdrx_fdecim_save_stages(m_last, m_stages) != SDRX_OK) || sdrx_fdecim_load_stages(h, m_stages) != SDRX_OK) {
                std::fprintf(stderr, "sdrx float decimators: %s\n", sdrx_last_error()); return;
            }
            m_last = h;
        }
        int32_t cnt = 0;
        if (sdrx_fdecim_process(h, buf, n, &**it, &cnt) != SDRX_OK) { std::fprintf(stderr, "sdrx float decimators: %s\n", sdrx_last_error()); return; }
        *it += cnt;
    }
    int m_device;
    sdrx_fdecim_t* m_h[7][3];
    sdrx_fdecim_stages_t* m_stages;
    sdrx_fdecim_t* m_last;
};
// DecimatorsFI: float in, int16 Sample out (decimatorsfi.h:29-57; AirspyHF thread)
class DecimatorsFI : public FloatDecimatorsBase<SampleVector, float, SDRX_FD_IN_F32, SDRX_FD_OUT_I16, 16> {
public: explicit DecimatorsFI(int device = 0) : FloatDecimatorsBase(device) {}
};
// DecimatorsFF: float in, float FSample out (decimatorsff.h)
class DecimatorsFF : public FloatDecimatorsBase<FSampleVector, float, SDRX_FD_IN_F32, SDRX_FD_OUT_F32, 16> {
public: explicit DecimatorsFF(int device = 0) : FloatDecimatorsBase(device) {}
};
// DecimatorsIF<qint16, InputBits>: int16 in, float out scaled by decimation_scale<InputBits> (decimatorsif.h:52-79)
template<typename T, unsigned InputBits>
class DecimatorsIF : public FloatDecimatorsBase<FSampleVector, T, SDRX_FD_IN_I16, SDRX_FD_OUT_F32, (int)InputBits> {
    static_assert(sizeof(T) == 2, "this build covers DecimatorsIF<qint16, {8,12,16}>");
public: explicit DecimatorsIF(int device = 0) : FloatDecimatorsBase<FSampleVector, T, SDRX_FD_IN_I16, SDRX_FD_OUT_F32, (int)InputBits>(device) {}
};

// N DownChannelizers on one device stream.  configure() == DownChannelizer::configure(queue, rate, fc)
// (downchannelizer.cpp:44-48) for one channel; feed() == the engine feeding every channel's
// DownChannelizer::feed with the same span; pull() hands over what m_sampleSink->feed would have got.
class DownChannelizerBank {
public:
    DownChannelizerBank(int inputSampleRate, const std::vector<int32_t>& requestedRates,
                        const std::vector<int32_t>& requestedCenters, int device = 0) : m_h(nullptr)
    {
        if (sdrx_chan_bank_create(&m_h, device, inputSampleRate, (int32_t)requestedRates.size(),
                                  requestedRates.data(), requestedCenters.data()) != SDRX_OK)
            std::fprintf(stderr, "sdrx::DownChannelizerBank: %s\n", sdrx_last_error());
    }
    ~DownChannelizerBank() { if (m_h) sdrx_chan_bank_destroy(m_h); }
    DownChannelizerBank(const DownChannelizerBank&) = delete;
    DownChannelizerBank& operator=(const DownChannelizerBank&) = delete;
    bool ok() const { return m_h != nullptr; }

    void configure(int channel, int sampleRate, int centerFrequency) { sdrx_chan_bank_reconfigure(m_h, channel, sampleRate, centerFrequency); }
    // MsgChannelizerNotification contents (downchannelizer.cpp:184-187)
    int getOutputSampleRate(int channel) const { int32_t r = 0; sdrx_chan_bank_info(m_h, channel, nullptr, nullptr, &r, nullptr); return r; }
    int getFrequencyOffset(int channel) const { int32_t o = 0; sdrx_chan_bank_info(m_h, channel, nullptr, nullptr, nullptr, &o); return o; }

    void feed(const SampleVector::const_iterator& begin, const SampleVector::const_iterator& end, bool /*positiveOnly*/)
    {
        if (end == begin) return;
        if (sdrx_chan_bank_feed(m_h, reinterpret_cast<const int16_t*>(&*begin), (int64_t)(end - begin)) != SDRX_OK)
            std::fprintf(stderr, "sdrx::DownChannelizerBank::feed: %s\n", sdrx_last_error());
    }
    // appends channel's pending output (the m_sampleBuffer of downchannelizer.cpp:87) to `out`
    size_t pull(int channel, SampleVector& out)
    {
        const int64_t n = sdrx_chan_bank_available(m_h, channel);
        if (n <= 0) return 0;
        const size_t at = out.size();
        out.resize(at + (size_t)n);
        const int64_t got = sdrx_chan_bank_read(m_h, channel, reinterpret_cast<int16_t*>(&out[at]), n);
        out.resize(at + (size_t)(got > 0 ? got : 0));
        return (size_t)(got > 0 ? got : 0);
    }
private:
    sdrx_chan_bank_t* m_h;
};

class SampleSinkFifo {                             // dsp/samplesinkfifo.h:27-65
public:
    explicit SampleSinkFifo(int size = 0) : m_h(nullptr) { sdrx_fifo_create(&m_h, (uint32_t)size); }
    ~SampleSinkFifo() { sdrx_fifo_destroy(m_h); }
    SampleSinkFifo(const SampleSinkFifo&) = delete;
    SampleSinkFifo& operator=(const SampleSinkFifo&) = delete;
    bool setSize(int size) { return sdrx_fifo_set_size(m_h, (uint32_t)size) == SDRX_OK; }
    unsigned size() { return sdrx_fifo_size(m_h); }
    unsigned fill() { return sdrx_fifo_fill(m_h); }
    unsigned write(const uint8_t* data, unsigned count) { return sdrx_fifo_write_bytes(m_h, data, count); }
    unsigned write(SampleVector::const_iterator begin, SampleVector::const_iterator end)
    { return begin == end ? 0 : sdrx_fifo_write(m_h, reinterpret_cast<const int16_t*>(&*begin), (uint32_t)(end - begin)); }
    unsigned read(SampleVector::iterator begin, SampleVector::iterator end)
    { return begin == end ? 0 : sdrx_fifo_read(m_h, reinterpret_cast<int16_t*>(&*begin), (uint32_t)(end - begin)); }
    // spans as raw pointers into the ring (the reference returns vector iterators into m_data)
    unsigned readBegin(unsigned count, const Sample** part1Begin, const Sample** part1End,
                       const Sample** part2Begin, const Sample** part2End)
    {
        const int16_t *p1, *p2; uint32_t n1, n2;
        const unsigned tot = sdrx_fifo_read_begin(m_h, count, &p1, &n1, &p2, &n2);
        *part1Begin = reinterpret_cast<const Sample*>(p1); *part1End = *part1Begin + n1;
        *part2Begin = reinterpret_cast<const Sample*>(p2); *part2End = *part2Begin + n2;
        return tot;
    }
    unsigned readCommit(unsigned count) { return sdrx_fifo_read_commit(m_h, count); }
    void onDataReady(sdrx_fifo_data_ready_cb cb, void* user) { sdrx_fifo_on_data_ready(m_h, cb, user); }   // signal dataReady()
private:
    sdrx_fifo_t* m_h;
};

} // namespace sdrx
