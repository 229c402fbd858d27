// TEST INFRASTRUCTURE ONLY.  C-ABI wrapper around the reference's real DownChannelizer
// (QObject; sdrbase/dsp/downchannelizer.{h,cpp}) so that the float bisection
// (createFilterChain, downchannelizer.cpp:250-287) and feed() (:50-91) can be run as they are.
// Needs Qt5Core + moc, both present in this image under /opt/conda; built by `make ref_qt`
// into oracle/_ref/libsdrref_qt.so.  Never shipped, never loaded by the product.
#include <stdint.h>
#include <vector>
#include "dsp/downchannelizer.h"
#include "dsp/dspcommands.h"
#include "util/messagequeue.h"
#include "dsp/samplesinkfifo.h"
#include "dsp/filerecord.h"
#include <fstream>

namespace {

class CollectSink : public BasebandSampleSink {
public:
    std::vector<Sample> got;
    void start() override {}
    void stop() override {}
    void feed(const SampleVector::const_iterator& b, const SampleVector::const_iterator& e, bool) override
    { got.insert(got.end(), b, e); }
    bool handleMessage(const Message&) override { return false; }
};

class OpenChannelizer : public DownChannelizer {
public:
    explicit OpenChannelizer(BasebandSampleSink* s) : DownChannelizer(s) {}
    int nStages() const { return (int) m_filterStages.size(); }
    void modes(uint8_t* out) const
    {
        int i = 0;
        for (FilterStages::const_iterator it = m_filterStages.begin(); it != m_filterStages.end(); ++it)
            out[i++] = (uint8_t) (*it)->m_mode;     // ModeCenter=0, ModeLowerHalf=1, ModeUpperHalf=2
    }
    int outRate() const { return m_currentOutputSampleRate; }
    int ofs() const { return m_currentCenterFrequency; }
};

struct Holder {
    CollectSink sink;
    OpenChannelizer chan;
    Holder() : chan(&sink) {}
};

void drain(MessageQueue* q) { Message* m; while ((m = q->pop()) != 0) delete m; }

} // namespace

extern "C" {

void* refqt_chan_new(int in_rate, int req_rate, int req_fc)
{
    Holder* h = new Holder;
    DSPSignalNotification sig(in_rate, 0);
    h->chan.handleMessage(sig);                       // sets m_inputSampleRate, applyConfiguration
    DSPConfigureChannelizer cfg(req_rate, req_fc);
    h->chan.handleMessage(cfg);                       // the message DownChannelizer::configure posts
    drain(h->sink.getInputMessageQueue());
    return h;
}
void refqt_chan_free(void* p) { Holder* h = static_cast<Holder*>(p); drain(h->sink.getInputMessageQueue()); delete h; }

int refqt_chan_plan(void* p, uint8_t* modes, int* out_rate, int* ofs)
{
    Holder* h = static_cast<Holder*>(p);
    h->chan.modes(modes); *out_rate = h->chan.outRate(); *ofs = h->chan.ofs();
    return h->chan.nStages();
}

int64_t refqt_chan_feed(void* p, const int16_t* iq, int64_t n_cplx, int16_t* out)
{
    Holder* h = static_cast<Holder*>(p);
    SampleVector v((size_t) n_cplx);
    for (int64_t i = 0; i < n_cplx; i++) v[i] = Sample(iq[2*i], iq[2*i+1]);
    h->sink.got.clear();
    h->chan.feed(v.begin(), v.end(), false);
    for (size_t i = 0; i < h->sink.got.size(); i++) { out[2*i] = h->sink.got[i].real(); out[2*i+1] = h->sink.got[i].imag(); }
    return (int64_t) h->sink.got.size();
}

} // extern "C"


// The real SampleSinkFifo (sdrbase/dsp/samplesinkfifo.{h,cpp}), for pinning the sdrx_fifo_* mirror: write / fill /
// readBegin (two spans, reported as offsets into the ring and lengths) / readCommit / read.
extern "C" {
void* refqt_fifo_new(int size) { return new SampleSinkFifo(size); }
void refqt_fifo_free(void* p) { delete static_cast<SampleSinkFifo*>(p); }
unsigned refqt_fifo_fill(void* p) { return static_cast<SampleSinkFifo*>(p)->fill(); }
unsigned refqt_fifo_write(void* p, const int16_t* iq, unsigned n_cplx)
{
    SampleVector v(n_cplx);
    for (unsigned i = 0; i < n_cplx; i++) v[i] = Sample(iq[2*i], iq[2*i+1]);
    return static_cast<SampleSinkFifo*>(p)->write(v.begin(), v.end());
}
unsigned refqt_fifo_write_bytes(void* p, const uint8_t* data, unsigned n_bytes) { return static_cast<SampleSinkFifo*>(p)->write(data, n_bytes); }
unsigned refqt_fifo_read(void* p, int16_t* out, unsigned n_cplx)
{
    SampleVector v(n_cplx);
    const unsigned n = static_cast<SampleSinkFifo*>(p)->read(v.begin(), v.end());
    for (unsigned i = 0; i < n; i++) { out[2*i] = v[i].real(); out[2*i+1] = v[i].imag(); }
    return n;
}
// copies what the two spans hold into out (span 1 then span 2); returns the total, *n1 / *n2 the span lengths
unsigned refqt_fifo_read_begin(void* p, unsigned count, int16_t* out, unsigned* n1, unsigned* n2)
{
    SampleVector::iterator b1, e1, b2, e2;
    const unsigned tot = static_cast<SampleSinkFifo*>(p)->readBegin(count, &b1, &e1, &b2, &e2);
    *n1 = (unsigned)(e1 - b1); *n2 = (unsigned)(e2 - b2);
    unsigned k = 0;
    for (SampleVector::iterator it = b1; it != e1; ++it, ++k) { out[2*k] = it->real(); out[2*k+1] = it->imag(); }
    for (SampleVector::iterator it = b2; it != e2; ++it, ++k) { out[2*k] = it->real(); out[2*k+1] = it->imag(); }
    return tot;
}
unsigned refqt_fifo_read_commit(void* p, unsigned count) { return static_cast<SampleSinkFifo*>(p)->readCommit(count); }
}


// The real FileRecord (sdrbase/dsp/filerecord.{h,cpp}) as the .sdriq writer / header reader, for pinning sdrx_sdriq_*:
// the sink is told the rate and centre frequency the way the engine tells it (DSPSignalNotification), records one feed.
extern "C" {
int refqt_filerecord_write(const char* path, int rate, long long centre, const int16_t* iq, unsigned n_cplx)
{
    FileRecord fr;
    fr.setFileName(QString::fromUtf8(path));
    DSPSignalNotification sig(rate, centre);
    fr.handleMessage(sig);
    fr.startRecording();
    SampleVector v(n_cplx);
    for (unsigned i = 0; i < n_cplx; i++) v[i] = Sample(iq[2*i], iq[2*i+1]);
    fr.feed(v.begin(), v.end(), false);
    fr.stopRecording();
    return 0;
}
int refqt_filerecord_read_header(const char* path, int* rate, unsigned long long* centre, long long* ts, unsigned* sample_size)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return -1;
    FileRecord::Header h;
    FileRecord::readHeader(f, h);
    *rate = h.sampleRate; *centre = h.centerFrequency; *ts = (long long) h.startTimeStamp; *sample_size = h.sampleSize;
    return 0;
}
}
