// Qt adapter: a FileSource-shaped DeviceSampleSource -- .sdriq replay into the device FIFO, the way
// plugins/samplesource/filesource/filesourcethread.cpp:170-251 does it (chunk = rate * throttle, read, SampleSinkFifo::write
// of the raw bytes, 24-bit files narrowed by >> 8 for the 16-bit build, end of file: rewind to sizeof(FileRecord::Header) --
// 32 with struct padding, so loop playback skips the first two samples: the reference's quirk, kept).  The reference paces
// tick() with a QTimer; here the owner calls tick(ms) itself (a test drives it deterministically, a replay tool from a
// timer), so N device sets can be pumped as fast as the GPU engine drains them.
#ifndef SDRX_QT_FILESOURCEPUMP_H
#define SDRX_QT_FILESOURCEPUMP_H

#include <fstream>
#include <vector>
#include <QString>
#include "dsp/devicesamplesource.h"
#include "sdrx.h"

class FileSourcePump : public DeviceSampleSource {
public:
    explicit FileSourcePump(const QString& fileName);
    virtual ~FileSourcePump();
    virtual void destroy() { delete this; }
    virtual void init() {}
    virtual bool start();                                  //!< opens the file, reads the header (FileSourceInput::openFileStream)
    virtual void stop();
    virtual QByteArray serialize() const { return QByteArray(); }
    virtual bool deserialize(const QByteArray&) { return true; }
    virtual const QString& getDeviceDescription() const { return m_deviceDescription; }
    virtual int getSampleRate() const { return m_header.sample_rate; }
    virtual quint64 getCenterFrequency() const { return m_header.center_frequency; }
    virtual void setCenterFrequency(qint64) {}
    virtual bool handleMessage(const Message&) { return false; }
    virtual void setMessageQueueToGUI(MessageQueue* queue) { m_guiMessageQueue = queue; }

    bool readHeader();                                     //!< header only (rate / centre frequency known before start())
    unsigned int tick(int throttleMs);                     //!< one FileSourceThread::tick(): returns the samples written to the FIFO
    quint64 getSamplesCount() const { return m_samplesCount; }

private:
    QString m_fileName, m_deviceDescription;
    std::ifstream m_ifstream;
    sdrx_sdriq_header m_header;
    std::vector<quint8> m_fileBuf, m_convertBuf;
    quint64 m_samplesCount;
    bool m_running;
};

#endif
