// TEST INFRASTRUCTURE ONLY.  A device thread of the reference, source UNCHANGED (plugins/samplesource/testsource/
// testsourcethread.{h,cpp}: synthetic carrier -> Decimators<qint32,qint16,SDR_RX_SAMP_SZ,{8,12,16}> -> SampleSinkFifo::write), built
// twice from this file: against the reference's dsp/decimators.h (oracle/_ref/testsource_ref) and against the shadow headers of
// qt_adapter/shadow (oracle/_ref/testsource_gpu: the thread's three Decimators members are sdrx::Decimators).  Both run the
// thread off a QTimer for a while and dump what arrived in the (reference) SampleSinkFifo; the test compares the dumps.
// The sample rate is chosen so that every tick's chunk is a whole number of decimation groups whatever the timer did
// (2.56 MS/s: 2560 samples per elapsed ms), so the stream does not depend on timing -- only its length does.
//   usage: <exe> <log2Decim> <fcPos> <bitSizeIndex> <out_file> [ticks]
#include <QCoreApplication>
#include <QTimer>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "testsourcethread.h"
#include "dsp/samplesinkfifo.h"

int main(int argc, char* argv[])
{
    QCoreApplication app(argc, argv);
    if (argc < 5) { fprintf(stderr, "usage: %s log2 fcpos bits out [ticks]\n", argv[0]); return 2; }
    const int log2 = atoi(argv[1]), fcpos = atoi(argv[2]), bits = atoi(argv[3]);
    const int ticks = argc > 5 ? atoi(argv[5]) : 12;
    SampleSinkFifo fifo(1 << 22);
    TestSourceThread thread(&fifo);
    thread.setSamplerate(2560000);
    thread.setLog2Decimation((unsigned) log2);
    thread.setFcPos(fcpos);
    thread.setBitSize((uint32_t) bits);
    thread.setAmplitudeBits(bits == 0 ? 100 : bits == 1 ? 1800 : 30000);
    thread.setDCFactor(0.03f); thread.setIFactor(0.02f); thread.setQFactor(-0.015f); thread.setPhaseImbalance(0.01f);
    thread.setFrequencyShift(123456);
    thread.setToneFrequency(1000);
    thread.setModulation(TestSourceSettings::ModulationAM);
    thread.setAMModulation(0.5f);
    QTimer timer;
    thread.connectTimer(timer);
    int n = 0;
    QObject::connect(&timer, &QTimer::timeout, [&] { if (++n >= ticks) { timer.stop(); app.quit(); } });
    thread.startWork();
    timer.start(7);
    app.exec();
    thread.stopWork();
    std::vector<Sample> out(fifo.fill());
    const unsigned got = fifo.read(out.begin(), out.end());
    FILE* f = fopen(argv[4], "wb");
    if (!f) return 3;
    fwrite(out.data(), sizeof(Sample), got, f);
    fclose(f);
    printf("%u samples from the FIFO after %d ticks\n", got, n);
    return got > 0 ? 0 : 4;
}
