// TEST INFRASTRUCTURE ONLY.  Drop-in check at the reference's own C++ interfaces, in one process:
//
//   (1) consumer side: N reference DownChannelizer objects, each feeding a recording
//       BasebandSampleSink, next to ONE GpuDownChannelizerBank (qt_adapter/) feeding N more recording sinks.
//       Both sides get the same DSPSignalNotification / channel configuration and the same SampleVector
//       spans through BasebandSampleSink::feed(); the recorded samples and the MsgChannelizerNotification
//       (rate, offset) each demod would have received must be identical.  One channel is re-configured
//       in the middle of the stream.
//   (3) the chain device thread -> Decimators -> SampleSinkFifo -> a DSPDeviceSourceEngine::work-shaped drain loop ->
//       channelizers -> demod sinks, reference objects on one side, sdrx::Decimators + the GPU bank on the other.
//   (2) producer side: the reference's Decimators<qint32,qint16,SDR_RX_SAMP_SZ,12> next to
//       sdrx::Decimators<...> (include/sdrx/dsp.hpp) with the device thread's call pattern
//       (limesdrinputthread.cpp:103-135): decimateK_x(&it, buf, len) into a SampleVector, block by block.
//
// Built here by `make -C oracle dropin` from the reference sources where they lie (+ moc + Qt5Core of the
// image) into oracle/_ref/dropin_test; run on the GPU box by tests/test_dropin_gpu.py.  Prints one line per
// check and exits non-zero on any mismatch.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "dsp/downchannelizer.h"
#include "dsp/dspcommands.h"
#include "dsp/decimators.h"
#include "util/messagequeue.h"
#include "dsp/samplesinkfifo.h"
#include "util/movingaverage.h"
#include "gpudownchannelizerbank.h"
#define SDRX_HOST_SAMPLE ::Sample      // sdrx::Decimators then takes the reference's SampleVector::iterator*
#include "sdrx/dsp.hpp"
#include "dropin_common.hpp"

uint32_t dropin_rng_state = 12345u;
int dropin_fails = 0;
void producer_side_u(int device);      // dropin_test_u.cpp
void producer_side_f(int device);      // dropin_test_f.cpp

namespace {

class RecorderSink : public BasebandSampleSink {
public:
    std::vector<Sample> got;
    std::vector<int> notes;            // (rate, offset) pairs of MsgChannelizerNotification, in arrival order
    int feeds;
    RecorderSink() : feeds(0) {}
    void start() override {}
    void stop() override {}
    void feed(const SampleVector::const_iterator& b, const SampleVector::const_iterator& e, bool) override
    { got.insert(got.end(), b, e); feeds++; }
    // BasebandSampleSink::handleInputMessages (basebandsamplesink.cpp:19-25) runs on messageEnqueued -- a direct call here,
    // the demod's own thread in the application -- and deletes what handleMessage() accepts
    bool handleMessage(const Message& m) override
    {
        if (DownChannelizer::MsgChannelizerNotification::match(m)) {
            const DownChannelizer::MsgChannelizerNotification& n = (const DownChannelizer::MsgChannelizerNotification&) m;
            notes.push_back(n.getSampleRate()); notes.push_back(n.getFrequencyOffset());
        }
        return true;
    }
    void pump() { Message* m; while ((m = getInputMessageQueue()->pop()) != 0) { handleMessage(*m); delete m; } }
};

void consumer_side(int device)
{
    const int fs = 61440000, N = 12;
    std::vector<RecorderSink*> refSinks, gpuSinks;
    std::vector<DownChannelizer*> refs;
    GpuDownChannelizerBank bank(device);
    DSPSignalNotification sig(fs, 0);
    for (int c = 0; c < N; c++) {
        refSinks.push_back(new RecorderSink); gpuSinks.push_back(new RecorderSink);
        refs.push_back(new DownChannelizer(refSinks[c]));
        bank.addChannel(gpuSinks[c]);
    }
    for (int c = 0; c < N; c++) refs[c]->handleMessage(sig);
    bank.handleMessage(sig);
    std::vector<int> rate(N), fc(N);
    for (int c = 0; c < N; c++) {
        rate[c] = (c % 3 == 2) ? 96000 : (c % 4 == 1) ? 12000 : 48000;
        fc[c] = -15000000 + (int)(c * (30000000.0 / 11)) + 137 * c;      // mixes lower/upper/centre paths and depths 8..12
        DSPConfigureChannelizer cfg(rate[c], fc[c]);
        refs[c]->handleMessage(cfg);
        bank.configureChannel(c, rate[c], fc[c]);
    }
    // spans of uneven length, as the engine's work() hands them over (two parts of the FIFO ring)
    const int spans[] = { 32768, 1, 65535, 100000, 3, 262144, 77777 };
    long total = 0;
    for (size_t s = 0; s < sizeof spans / sizeof spans[0]; s++) {
        SampleVector v((size_t) spans[s]);
        for (int i = 0; i < spans[s]; i++) v[i] = Sample((qint16)((int)(rng() % 4096) - 2048), (qint16)((int)(rng() % 4096) - 2048));
        if (s == 4) {                  // DSPConfigureChannelizer for one channel while the others keep running
            rate[5] = 24000; fc[5] = 1234567;
            DSPConfigureChannelizer cfg(rate[5], fc[5]);
            refs[5]->handleMessage(cfg);
            bank.configureChannel(5, rate[5], fc[5]);
        }
        for (int c = 0; c < N; c++) refs[c]->feed(v.begin(), v.end(), false);
        bank.feed(v.begin(), v.end(), false);
        total += spans[s];
    }
    for (int c = 0; c < N; c++) {
        refSinks[c]->pump(); gpuSinks[c]->pump();
        char what[96];
        const std::vector<Sample>& a = refSinks[c]->got; const std::vector<Sample>& b = gpuSinks[c]->got;
        bool same = a.size() == b.size();
        for (size_t i = 0; same && i < a.size(); i++) same = a[i].real() == b[i].real() && a[i].imag() == b[i].imag();
        snprintf(what, sizeof what, "channel %2d samples DownChannelizer vs GPU bank", c);
        report(what, same && !a.empty(), (long) a.size());
        // the last notification is the configuration in force; the reference posts one per applyConfiguration
        const std::vector<int>& na = refSinks[c]->notes; const std::vector<int>& nb = gpuSinks[c]->notes;
        const bool notes_ok = na.size() >= 2 && nb.size() >= 2 && na[na.size() - 2] == nb[nb.size() - 2] && na[na.size() - 1] == nb[nb.size() - 1];
        snprintf(what, sizeof what, "channel %2d MsgChannelizerNotification (rate %d, ofs %d)", c, na.size() >= 2 ? na[na.size() - 2] : -1, na.size() >= 2 ? na[na.size() - 1] : -1);
        report(what, notes_ok, (long) nb.size() / 2);
        if (!notes_ok) {
            printf("    reference:"); for (size_t i = 0; i < na.size(); i++) printf(" %d", na[i]);
            printf("\n    gpu bank: "); for (size_t i = 0; i < nb.size(); i++) printf(" %d", nb[i]);
            printf("\n");
        }
    }
    printf("consumer side: %ld input samples through %d channels\n", total, N);
    for (int c = 0; c < N; c++) { delete refs[c]; refSinks[c]->pump(); gpuSinks[c]->pump(); }
}

void producer_side(int device)
{
    typedef Decimators<qint32, qint16, SDR_RX_SAMP_SZ, 12> RefDec12;
    typedef sdrx::Decimators<qint32, qint16, SDR_RX_SAMP_SZ, 12> GpuDec12;
    typedef Decimators<qint32, qint16, SDR_RX_SAMP_SZ, 16> RefDec16;
    typedef sdrx::Decimators<qint32, qint16, SDR_RX_SAMP_SZ, 16> GpuDec16;
    PRODUCER(RefDec12, GpuDec12, qint16, decimate64_cen, -2048, 4096)      // LimeSDR / sdrbench configuration
    PRODUCER(RefDec12, GpuDec12, qint16, decimate16_cen, -2048, 4096)
    PRODUCER(RefDec12, GpuDec12, qint16, decimate8_inf, -2048, 4096)
    PRODUCER(RefDec12, GpuDec12, qint16, decimate32_sup, -2048, 4096)
    PRODUCER(RefDec12, GpuDec12, qint16, decimate2_cen, -2048, 4096)
    PRODUCER(RefDec12, GpuDec12, qint16, decimate1, -2048, 4096)
    PRODUCER(RefDec16, GpuDec16, qint16, decimate4_inf, -32768, 65536)      // full-scale 16-bit source
    {
        // ONE object, the device thread changes log2Decim / fcPos while running (limesdrinputthread.cpp:103-135 switches on
        // m_log2Decim and m_fcPos in every callback): all cascades share the object's six filters (decimators.h:326-333)
        RefDec12 ref; GpuDec12 gpu(device);
        int call = 0;
        auto both = [&](SampleVector::iterator* itR, SampleVector::iterator* itG, const qint16* b, qint32 len) {
            switch (call) {
            case 0: ref.decimate64_cen(itR, b, len); gpu.decimate64_cen(itG, b, len); break;
            case 1: ref.decimate8_inf(itR, b, len);  gpu.decimate8_inf(itG, b, len);  break;     // a short stay: 1000 samples
            case 2: ref.decimate64_cen(itR, b, len); gpu.decimate64_cen(itG, b, len); break;     // back: stages 4-6 still hold call 0's tail
            case 3: ref.decimate2_sup(itR, b, len);  gpu.decimate2_sup(itG, b, len);  break;
            case 4: ref.decimate1(itR, b, len);      gpu.decimate1(itG, b, len);      break;     // touches no filter
            case 5: ref.decimate32_sup(itR, b, len); gpu.decimate32_sup(itG, b, len); break;
            default: ref.decimate16_cen(itR, b, len); gpu.decimate16_cen(itG, b, len); break;
            }
        };
        const int lens[] = { 65536, 2000, 6000, 4096, 512, 30000, 65536, 65536 };
        SampleVector refOut(400000), gpuOut(400000);
        SampleVector::iterator itR = refOut.begin(), itG = gpuOut.begin();
        std::vector<qint16> buf(65536);
        for (int b = 0; b < 8; b++) {
            call = b < 6 ? b : 6;
            for (int i = 0; i < lens[b]; i++) buf[i] = (qint16)((int)(rng() % 4096) - 2048);
            both(&itR, &itG, buf.data(), lens[b]);
        }
        bool same = (itR - refOut.begin()) == (itG - gpuOut.begin());
        const long n = (long)(itR - refOut.begin());
        for (long i = 0; same && i < n; i++) same = refOut[i].real() == gpuOut[i].real() && refOut[i].imag() == gpuOut[i].imag();
        report("RefDec12: one object, K / fcPos changed at run time", same && n > 0, n);
    }
}

// (3) the whole RX chain as the application runs it: device-thread blocks -> Decimators::decimate8_cen -> the reference's
// SampleSinkFifo::write -> a drain loop shaped like DSPDeviceSourceEngine::work (dspdevicesourceengine.cpp:325-408:
// readBegin -> up to two spans -> every sink's feed() -> readCommit) -> channelizers -> demod sinks.  Left: reference
// objects only.  Right: sdrx::Decimators + GpuDownChannelizerBank behind the same FIFO class and the same loop.
void engine_chain(int device)
{
    const int fsDev = 8 * 2400000, fs = 2400000, N = 6;
    Decimators<qint32, qint16, SDR_RX_SAMP_SZ, 12> refDec;
    sdrx::Decimators<qint32, qint16, SDR_RX_SAMP_SZ, 12> gpuDec(device);
    SampleSinkFifo refFifo(fs / 4), gpuFifo(fs / 4);      // small on purpose: the ring wraps, work() sees two-part reads
    std::vector<RecorderSink*> refSinks, gpuSinks;
    std::vector<DownChannelizer*> refs;
    GpuDownChannelizerBank bank(device);
    for (int c = 0; c < N; c++) {
        refSinks.push_back(new RecorderSink); gpuSinks.push_back(new RecorderSink);
        refs.push_back(new DownChannelizer(refSinks[c])); bank.addChannel(gpuSinks[c]);
    }
    DSPSignalNotification sig(fs, 435000000);
    for (int c = 0; c < N; c++) refs[c]->handleMessage(sig);
    bank.handleMessage(sig);
    for (int c = 0; c < N; c++) {
        const int rate = c == 3 ? 96000 : 48000, fc = -1000000 + c * 400000 + 1234;
        DSPConfigureChannelizer cfg(rate, fc);
        refs[c]->handleMessage(cfg); bank.configureChannel(c, rate, fc);
    }
    (void) fsDev;
    // m_dcOffsetCorrection on: work() corrects each span in place before the sinks see it (:339-343, 375-379).
    // Reference side: iqCorrections(begin, end, false) on the engine's own MovingAverageUtil members (:175-181, 255-259);
    // GPU side: sdrx_dccorr_process on the same span.
    struct Drain {
        MovingAverageUtil<int32_t, int64_t, 1024> m_iBeta, m_qBeta;
        sdrx_dccorr_t* gpu;
        Drain() : gpu(0) {}
        void correct(SampleVector::iterator b, SampleVector::iterator e)
        {
            if (gpu) { sdrx_dccorr_process(gpu, reinterpret_cast<int16_t*>(&*b), (int64_t)(e - b)); return; }
            for (SampleVector::iterator it = b; it < e; it++) {
                m_iBeta(it->real()); m_qBeta(it->imag());
                it->m_real -= (int32_t) m_iBeta; it->m_imag -= (int32_t) m_qBeta;
            }
        }
        void work(SampleSinkFifo& fifo, std::vector<BasebandSampleSink*>& sinks)
        {
            while (fifo.fill() > 0) {
                SampleVector::iterator p1b, p1e, p2b, p2e;
                const uint count = fifo.readBegin(fifo.fill(), &p1b, &p1e, &p2b, &p2e);
                if (p1b != p1e) { correct(p1b, p1e); for (size_t i = 0; i < sinks.size(); i++) sinks[i]->feed(p1b, p1e, false); }
                if (p2b != p2e) { correct(p2b, p2e); for (size_t i = 0; i < sinks.size(); i++) sinks[i]->feed(p2b, p2e, false); }
                fifo.readCommit(count);
            }
        }
    };
    Drain refDrain, gpuDrain;
    sdrx_dccorr_create(&gpuDrain.gpu, device);
    std::vector<BasebandSampleSink*> refEngineSinks(refs.begin(), refs.end()), gpuEngineSinks(1, &bank);
    const int block = 2 * 131072;                           // int16 per device callback
    std::vector<qint16> buf((size_t) block);
    SampleVector convR((size_t) block / 2), convG((size_t) block / 2);
    long fed = 0;
    for (int b = 0; b < 40; b++) {
        const int len = (b % 5 == 2) ? block - 14 : block;
        for (int i = 0; i < len; i++) buf[i] = (qint16)((int)(rng() % 3600) - 1800 + ((i & 1) ? -230 : 170));   // a DC offset to remove
        SampleVector::iterator itR = convR.begin(), itG = convG.begin();
        refDec.decimate8_cen(&itR, buf.data(), len);        // the device thread's callback (limesdrinputthread.cpp:103-135)
        gpuDec.decimate8_cen(&itG, buf.data(), len);
        refFifo.write(convR.begin(), itR);
        gpuFifo.write(convG.begin(), itG);
        fed += itR - convR.begin();
        if (b % 3 != 1) { refDrain.work(refFifo, refEngineSinks); gpuDrain.work(gpuFifo, gpuEngineSinks); }   // sometimes two blocks pile up
    }
    refDrain.work(refFifo, refEngineSinks); gpuDrain.work(gpuFifo, gpuEngineSinks);
    sdrx_dccorr_destroy(gpuDrain.gpu);
    for (int c = 0; c < N; c++) {
        char what[96];
        const std::vector<Sample>& a = refSinks[c]->got; const std::vector<Sample>& g = gpuSinks[c]->got;
        bool same = a.size() == g.size() && !a.empty();
        for (size_t i = 0; same && i < a.size(); i++) same = a[i].real() == g[i].real() && a[i].imag() == g[i].imag();
        snprintf(what, sizeof what, "engine chain: decimate8_cen -> FIFO -> work() + DC corr -> channel %d", c);
        report(what, same, (long) a.size());
    }
    printf("engine chain: %ld samples through the FIFO\n", fed);
    for (int c = 0; c < N; c++) delete refs[c];
}

} // namespace

int main(int argc, char** argv)
{
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    consumer_side(device);
    producer_side(device);
    engine_chain(device);
    producer_side_u(device);
    producer_side_f(device);
    printf(dropin_fails ? "DROP-IN CHECK FAILED: %d mismatches\n" : "DROP-IN CHECK PASSED%.0d\n", dropin_fails);
    return dropin_fails ? 1 : 0;
}
