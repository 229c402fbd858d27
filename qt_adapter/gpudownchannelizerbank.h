// Qt adapter: the GPU channelizer bank presented to SDRangel as ONE BasebandSampleSink that replaces N
// { ThreadedBasebandSampleSink -> DownChannelizer } pairs of a device set (SURVEY.md §8b "Consumer side",
// INTEGRATION.md §4).  Compiles against the reference's own headers (sdrbase/dsp/basebandsamplesink.h,
// dspcommands.h, downchannelizer.h for the notification message type) and links libsdrx.so.
//
//   engine->addSink(bank)                       // direct sink: gets the device FIFO spans in work()
//   int ch = bank->addChannel(demod)             // demod = the plugin's BasebandSampleSink (NFMDemod, SSBDemod, ...)
//   bank->configureChannel(ch, 48000, fc)        // what DownChannelizer::configure posted per channel
//
// feed() hands the span to sdrx_chan_bank_feed once, then gives every demod exactly the samples its own
// DownChannelizer would have produced (bit-identical), so the demods are unchanged.
#ifndef SDRX_QT_GPUDOWNCHANNELIZERBANK_H
#define SDRX_QT_GPUDOWNCHANNELIZERBANK_H

#include <vector>
#include "dsp/basebandsamplesink.h"
#include "sdrx.h"

class GpuDownChannelizerBank : public BasebandSampleSink {
    Q_OBJECT
public:
    explicit GpuDownChannelizerBank(int device = 0);
    virtual ~GpuDownChannelizerBank();

    int addChannel(BasebandSampleSink* demod);                       //!< returns the channel index
    void configureChannel(int channel, int sampleRate, int centerFrequency);
    int getInputSampleRate() const { return m_inputSampleRate; }

    virtual void start();
    virtual void stop();
    virtual void feed(const SampleVector::const_iterator& begin, const SampleVector::const_iterator& end, bool positiveOnly);
    virtual bool handleMessage(const Message& cmd);

private:
    struct Channel { BasebandSampleSink* sink; int reqRate; int reqFc; };
    void rebuild();                                                  //!< (re)create the bank from the current configuration
    void notify(int channel);                                        //!< push MsgChannelizerNotification to the demod

    int m_device;
    int m_inputSampleRate;
    std::vector<Channel> m_channels;
    sdrx_chan_bank_t* m_bank;
    SampleVector m_scratch;
};

#endif
