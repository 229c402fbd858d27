#include "gpudownchannelizerbank.h"
#include "dsp/dspcommands.h"
#include "dsp/downchannelizer.h"      // DownChannelizer::MsgChannelizerNotification: the message demods already handle
#include "util/messagequeue.h"
#include <QDebug>

GpuDownChannelizerBank::GpuDownChannelizerBank(int device) :
    m_device(device), m_inputSampleRate(0), m_bank(0)
{
    setObjectName("GpuDownChannelizerBank");
}

GpuDownChannelizerBank::~GpuDownChannelizerBank()
{
    if (m_bank) sdrx_chan_bank_destroy(m_bank);
}

int GpuDownChannelizerBank::addChannel(BasebandSampleSink* demod)
{
    Channel c; c.sink = demod; c.reqRate = 48000; c.reqFc = 0;
    m_channels.push_back(c);
    const int index = (int) m_channels.size() - 1;
    if (m_bank) {
        // a DownChannelizer added next to running ones does not disturb them (DeviceSourceAPI::addThreadedSink):
        // the bank grows by one chain that starts from zero history; the others keep history and queued output
        int32_t got = -1;
        if (sdrx_chan_bank_add_channel(m_bank, c.reqRate, c.reqFc, &got) != SDRX_OK || got != index) {
            qCritical("GpuDownChannelizerBank::addChannel: %s", sdrx_last_error());
            rebuild();
        } else {
            notify(index);
        }
    } else {
        rebuild();                                                 // first channel, or no input rate yet
    }
    return index;
}

void GpuDownChannelizerBank::configureChannel(int channel, int sampleRate, int centerFrequency)
{
    // DownChannelizer::handleMessage(DSPConfigureChannelizer) -> applyConfiguration (downchannelizer.cpp:131-144,157-189)
    m_channels[channel].reqRate = sampleRate;
    m_channels[channel].reqFc = centerFrequency;
    if (m_bank) {
        // one channel only: its chain restarts from zero history, the others keep running (:167-171)
        if (sdrx_chan_bank_reconfigure(m_bank, channel, sampleRate, centerFrequency) != SDRX_OK)
            qCritical("GpuDownChannelizerBank: %s", sdrx_last_error());
        notify(channel);
    }
}

void GpuDownChannelizerBank::rebuild()
{
    if (m_bank) { sdrx_chan_bank_destroy(m_bank); m_bank = 0; }
    if (m_inputSampleRate == 0 || m_channels.empty()) return;      // "m_inputSampleRate=0 aborting"
    std::vector<int32_t> rates, fcs;
    for (size_t i = 0; i < m_channels.size(); i++) { rates.push_back(m_channels[i].reqRate); fcs.push_back(m_channels[i].reqFc); }
    if (sdrx_chan_bank_create(&m_bank, m_device, m_inputSampleRate, (int32_t) rates.size(), rates.data(), fcs.data()) != SDRX_OK) {
        qCritical("GpuDownChannelizerBank: %s", sdrx_last_error());
        m_bank = 0;
        return;
    }
    for (size_t i = 0; i < m_channels.size(); i++) notify((int) i);
}

void GpuDownChannelizerBank::notify(int channel)
{
    int32_t outRate = 0, ofs = 0;
    sdrx_chan_bank_info(m_bank, channel, 0, 0, &outRate, &ofs);
    BasebandSampleSink* sink = m_channels[channel].sink;
    if (sink != 0) {
        // same message, same queue as DownChannelizer::applyConfiguration (:184-187)
        sink->getInputMessageQueue()->push(DownChannelizer::MsgChannelizerNotification::create(outRate, ofs));
    }
}

void GpuDownChannelizerBank::start()
{
    for (size_t i = 0; i < m_channels.size(); i++) if (m_channels[i].sink) m_channels[i].sink->start();
}

void GpuDownChannelizerBank::stop()
{
    for (size_t i = 0; i < m_channels.size(); i++) if (m_channels[i].sink) m_channels[i].sink->stop();
}

void GpuDownChannelizerBank::feed(const SampleVector::const_iterator& begin, const SampleVector::const_iterator& end, bool positiveOnly)
{
    if (!m_bank || begin == end) return;
    // Sample is a packed {qint16 re, im} (dsptypes.h:44-65): the vector storage IS the int16 I/Q stream
    if (sdrx_chan_bank_feed(m_bank, reinterpret_cast<const int16_t*>(&*begin), (int64_t)(end - begin)) != SDRX_OK) {
        qCritical("GpuDownChannelizerBank::feed: %s", sdrx_last_error());
        return;
    }
    for (size_t c = 0; c < m_channels.size(); c++) {
        const int64_t n = sdrx_chan_bank_available(m_bank, (int32_t) c);
        if (n <= 0 || m_channels[c].sink == 0) { if (n > 0) sdrx_chan_bank_skip(m_bank, (int32_t) c, -1); continue; }
        m_scratch.resize((size_t) n);
        const int64_t got = sdrx_chan_bank_read(m_bank, (int32_t) c, reinterpret_cast<int16_t*>(&m_scratch[0]), n);
        if (got > 0) m_channels[c].sink->feed(m_scratch.begin(), m_scratch.begin() + got, positiveOnly);   // downchannelizer.cpp:87
    }
}

bool GpuDownChannelizerBank::handleMessage(const Message& cmd)
{
    if (DSPSignalNotification::match(cmd)) {
        // engine broadcast at gotoInit (dspdevicesourceengine.cpp:455-515): new input rate -> every chain is re-planned
        const DSPSignalNotification& notif = (const DSPSignalNotification&) cmd;
        m_inputSampleRate = notif.getSampleRate();
        rebuild();
        for (size_t i = 0; i < m_channels.size(); i++) {
            if (m_channels[i].sink) m_channels[i].sink->getInputMessageQueue()->push(new DSPSignalNotification(notif));
        }
        return true;
    }
    return false;
}
