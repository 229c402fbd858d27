// TEST INFRASTRUCTURE (built into oracle/_ref/, run on the GPU box by tests/test_dropin_gpu.py).
// The REAL DSPDeviceSourceEngine (compiled from /root/reference/sdrbase/dsp/dspdevicesourceengine.cpp with the image's Qt)
// next to qt_adapter/GpuDeviceSourceEngine in one process: same DeviceSampleSource subclass, same block sequence written
// into the source FIFO (SampleSinkFifo::write -> dataReady() -> handleData() -> work()), the same sinks:
//   * a collector (what a spectrum / FileRecord sink sees) -- compared byte for byte, corrections off / DC / DC + I/Q imbalance,
//     with a configureCorrections() in the middle of the stream (every average restarts);
//   * real DownChannelizer objects as direct sinks of the real engine vs ONE GpuDownChannelizerBank on the GPU engine;
//   * the state machine: notStarted -> idle -> ready -> running -> idle, "No sample source configured" error.
#include <QCoreApplication>
#include <QThread>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>
#include "dsp/dspdevicesourceengine.h"
#include "dsp/devicesamplesource.h"
#include "dsp/basebandsamplesink.h"
#include "dsp/downchannelizer.h"
#include "dsp/dspcommands.h"
#include "gpudevicesourceengine.h"
#include "gpudownchannelizerbank.h"
#include "filesourcepump.h"
#include <fstream>

namespace {

struct TestSource : public DeviceSampleSource {                    // FileSource-shaped: a FIFO somebody writes blocks into
    QString m_desc; int m_rate; bool m_started;
    TestSource(int rate) : m_desc("sdrx test source"), m_rate(rate), m_started(false) { m_sampleFifo.setSize(1 << 22); }
    virtual void destroy() {}
    virtual void init() {}
    virtual bool start() { m_started = true; return true; }
    virtual void stop() { m_started = false; }
    virtual QByteArray serialize() const { return QByteArray(); }
    virtual bool deserialize(const QByteArray&) { return true; }
    virtual const QString& getDeviceDescription() const { return m_desc; }
    virtual int getSampleRate() const { return m_rate; }
    virtual quint64 getCenterFrequency() const { return 435000000ULL; }
    virtual void setCenterFrequency(qint64) {}
    virtual bool handleMessage(const Message&) { return false; }
    virtual void setMessageQueueToGUI(MessageQueue* q) { m_guiMessageQueue = q; }
};

struct Collector : public BasebandSampleSink {
    std::vector<Sample> got; int starts, stops, notifs; int lastRate;
    Collector() : starts(0), stops(0), notifs(0), lastRate(-1) {}
    virtual void start() { starts++; }
    virtual void stop() { stops++; }
    virtual void feed(const SampleVector::const_iterator& b, const SampleVector::const_iterator& e, bool) { got.insert(got.end(), b, e); }
    virtual bool handleMessage(const Message& m) {
        if (DSPSignalNotification::match(m)) { notifs++; lastRate = ((const DSPSignalNotification&) m).getSampleRate(); return true; }
        return false;
    }
};

std::vector<Sample> make_stream(size_t n, unsigned seed)
{
    std::vector<Sample> x(n);
    unsigned long long s = seed * 2654435761ULL + 12345;
    for (size_t i = 0; i < n; i++) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        const int a = (int)((s >> 33) % 16001) - 8000;
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        const int b = (int)((s >> 33) % 16001) - 8000;
        const int tone = (int)(6000.0 * ((i % 64) < 32 ? 1.0 : -1.0));
        int re = a + tone + 300, im = (int)(0.8 * b) + (int)(0.1 * a) - 200;     // DC + amplitude + phase imbalance
        x[i] = Sample((qint16) re, (qint16) im);
    }
    return x;
}

template<class Engine> void drain(Engine& e, TestSource& src)
{
    for (int i = 0; i < 20000 && src.getSampleFifo()->fill() > 0; i++) QThread::usleep(500);
    e.stopAcquistion();                                               // executes in the engine thread: the last work() has returned
}

int fails = 0;
void check(bool ok, const char* what) { if (!ok) { fails++; std::printf("FAIL: %s\n", what); } }

// one scenario on one engine type; returns what the collector saw
template<class Engine>
void scenario(Engine& eng, TestSource& src, Collector& col, const std::vector<Sample>& x, int mode, std::vector<BasebandSampleSink*> extra)
{
    eng.start();
    for (int i = 0; i < 2000 && eng.state() == Engine::StNotStarted; i++) QThread::usleep(500);
    check(eng.state() == Engine::StIdle, "idle after start");
    check(!eng.initAcquisition(), "initAcquisition without a source fails");
    check(eng.state() == Engine::StError && eng.errorMessage().contains("No sample source"), "error state + message without a source");
    eng.setSource(&src);
    eng.addSink(&col);
    for (size_t i = 0; i < extra.size(); i++) eng.addSink(extra[i]);
    check(eng.initAcquisition() && eng.state() == Engine::StReady, "ready after initAcquisition");
    check(col.lastRate == src.getSampleRate(), "sink got DSPSignalNotification(rate)");
    check(eng.startAcquisition() && eng.state() == Engine::StRunning && src.m_started && col.starts == 1, "running: source and sink started");
    if (mode) { eng.configureCorrections(true, mode == 2); QThread::msleep(20); }
    // ragged blocks through the FIFO, the way a device thread writes them
    const size_t cuts[] = { 0, 5, 4096, 4096 + 3, 70001, 300000, x.size() / 2, x.size() };
    for (size_t c = 0; c + 1 < sizeof cuts / sizeof cuts[0]; c++) {
        if (cuts[c] == x.size() / 2 && mode) {                       // mid-stream reconfigure: all averages restart
            for (int i = 0; i < 20000 && src.getSampleFifo()->fill() > 0; i++) QThread::usleep(500);
            QThread::msleep(30);
            eng.configureCorrections(true, mode == 2);
            QThread::msleep(30);
        }
        src.getSampleFifo()->write(x.begin() + cuts[c], x.begin() + cuts[c + 1]);
    }
    drain(eng, src);
    check(eng.state() == Engine::StIdle && !src.m_started && col.stops >= 1, "idle after stopAcquistion: source and sink stopped");
    eng.stop();
    eng.wait();
}

} // namespace

int main(int argc, char** argv)
{
    QCoreApplication app(argc, argv);
    const int rate = 2400000;
    const std::vector<Sample> x = make_stream(900000, 7);
    for (int mode = 0; mode < 3; mode++) {
        TestSource s1(rate), s2(rate);
        Collector c1, c2;
        std::vector<BasebandSampleSink*> e1, e2;
        // channel sinks: real DownChannelizers on the reference engine, one GPU bank on the GPU engine
        const int fcs[3] = { 0, 312500, -777000 };
        Collector d1[3], d2[3];
        DownChannelizer* dc[3];
        GpuDownChannelizerBank bank(0);
        if (mode == 0) {
            for (int k = 0; k < 3; k++) {
                dc[k] = new DownChannelizer(&d1[k]);
                { DSPConfigureChannelizer cfg(48000, fcs[k]); dc[k]->handleMessage(cfg); }   // what DownChannelizer::configure posts (downchannelizer.cpp:44-48)
                e1.push_back(dc[k]);
                const int ch = bank.addChannel(&d2[k]);
                bank.configureChannel(ch, 48000, fcs[k]);
            }
            e2.push_back(&bank);
        }
        {
            DSPDeviceSourceEngine ref(0);
            scenario(ref, s1, c1, x, mode, e1);
        }
        {
            GpuDeviceSourceEngine gpu(1, 0);
            scenario(gpu, s2, c2, x, mode, e2);
        }
        const bool same = c1.got.size() == c2.got.size() && c1.got.size() == x.size() &&
                          std::memcmp(&c1.got[0], &c2.got[0], c1.got.size() * sizeof(Sample)) == 0;
        std::printf("engine mode %d (%s): collector %zu vs %zu samples, %s\n", mode, mode == 0 ? "no correction" : mode == 1 ? "DC" : "DC + I/Q imbalance",
                    c1.got.size(), c2.got.size(), same ? "identical" : "DIFFERENT");
        check(same, "collector streams identical");
        if (mode == 0) {
            for (int k = 0; k < 3; k++) {
                const bool eq = d1[k].got.size() == d2[k].got.size() && !d1[k].got.empty() &&
                                std::memcmp(&d1[k].got[0], &d2[k].got[0], d1[k].got.size() * sizeof(Sample)) == 0;
                std::printf("  channel %d (fc %d): %zu vs %zu samples, %s\n", k, fcs[k], d1[k].got.size(), d2[k].got.size(), eq ? "identical" : "DIFFERENT");
                check(eq, "channel outputs identical");
            }
            for (int k = 0; k < 3; k++) delete dc[k];
        }
    }
    // ---- FileSource replay: .sdriq file -> FileSourcePump (DeviceSampleSource) -> engine -> sinks, on both engines;
    //      the file is shorter than what the ticks ask for, so the loop-rewind quirk (offset 32, two samples skipped) is crossed
    {
        const char* path = "/tmp/sdrx_dropin_engine.sdriq";
        const size_t nfile = 100003;
        {
            sdrx_sdriq_header hd; hd.sample_rate = 240000; hd.center_frequency = 145000000ULL; hd.start_timestamp = 1; hd.sample_size = 16;
            uint8_t raw[SDRX_SDRIQ_HEADER_BYTES]; sdrx_sdriq_write_header(raw, &hd);
            std::ofstream f(path, std::ios::binary); f.write(reinterpret_cast<const char*>(raw), sizeof raw);
            f.write(reinterpret_cast<const char*>(&x[0]), (std::streamsize)(nfile * sizeof(Sample)));
        }
        std::vector<Sample> seen[2];
        for (int which = 0; which < 2; which++) {
            FileSourcePump src(path);
            check(src.readHeader() && src.getSampleRate() == 240000, "sdriq header read before start");
            Collector col;
            auto pump = [&](auto& eng) {
                typedef typename std::remove_reference<decltype(eng)>::type E;
                eng.start();
                for (int i = 0; i < 2000 && eng.state() == E::StNotStarted; i++) QThread::usleep(500);
                eng.setSource(&src); eng.addSink(&col);
                check(eng.initAcquisition() && eng.startAcquisition(), "file source engine running");
                eng.configureCorrections(true, false); QThread::msleep(20);
                unsigned total = 0;
                for (int t = 0; t < 12; t++) {                                 // 12 ticks x 50 ms x 240 kS/s = 144000 samples > file
                    total += src.tick(50);
                    for (int i = 0; i < 20000 && src.getSampleFifo()->fill() > 0; i++) QThread::usleep(200);
                }
                eng.stopAcquistion(); eng.stop(); eng.wait();
                return total;
            };
            unsigned total;
            if (which == 0) { DSPDeviceSourceEngine e(2); total = pump(e); } else { GpuDeviceSourceEngine e(3, 0); total = pump(e); }
            check(total == col.got.size() && total == 12u * 12000u - 0u - (12u * 12000u > nfile ? (12000u - (unsigned)(nfile % 12000u)) : 0u), "pump wrote what the ticks asked for, short at end of file");
            seen[which] = col.got;
        }
        const bool same = seen[0].size() == seen[1].size() && !seen[0].empty() && std::memcmp(&seen[0][0], &seen[1][0], seen[0].size() * sizeof(Sample)) == 0;
        std::printf("file source replay (DC correction on, loop rewind crossed): %zu vs %zu samples, %s\n", seen[0].size(), seen[1].size(), same ? "identical" : "DIFFERENT");
        check(same, "file replay identical on both engines");
        std::remove(path);
    }
    std::printf(fails ? "ENGINE DROP-IN: %d FAILURES\n" : "ENGINE DROP-IN: ALL OK\n", fails);
    return fails ? 1 : 0;
}
