#include "gpudevicesourceengine.h"
#include "dsp/basebandsamplesink.h"
#include "dsp/devicesamplesource.h"
#include "dsp/threadedbasebandsamplesink.h"
#include "dsp/samplesinkfifo.h"
#include "dsp/dspcommands.h"
#include <QDebug>
#include <QMetaObject>

GpuDeviceSourceEngine::GpuDeviceSourceEngine(uint uid, int device, QObject* parent) :
    QThread(parent), m_uid(uid), m_device(device), m_state(StNotStarted), m_deviceSampleSource(0),
    m_sampleRate(0), m_centerFrequency(0), m_dcOffsetCorrection(false), m_iqImbalanceCorrection(false),
    m_dccorr(0), m_iqimb(0)
{
    moveToThread(this);                                 // like the reference: slots run in the engine's own thread
}

GpuDeviceSourceEngine::~GpuDeviceSourceEngine()
{
    stop();
    wait();
    if (m_dccorr) sdrx_dccorr_destroy(m_dccorr);
    if (m_iqimb) sdrx_iqimb_destroy(m_iqimb);
}

void GpuDeviceSourceEngine::run()
{
    m_state = StIdle;
    exec();
}

void GpuDeviceSourceEngine::start() { QThread::start(); }

void GpuDeviceSourceEngine::stop()
{
    if (isRunning()) call("cmdStop");
    m_state = StNotStarted;
    QThread::exit();
}

void GpuDeviceSourceEngine::call(const char* slot, void* arg, bool hasArg)
{
    // SyncMessenger::sendWait of the reference: the command executes in the engine thread, the caller waits
    const Qt::ConnectionType how = (QThread::currentThread() == this || !isRunning()) ? Qt::DirectConnection : Qt::BlockingQueuedConnection;
    if (hasArg) QMetaObject::invokeMethod(this, slot, how, Q_ARG(void*, arg));
    else QMetaObject::invokeMethod(this, slot, how);
}

bool GpuDeviceSourceEngine::initAcquisition() { call("cmdInit"); return m_state == StReady; }
bool GpuDeviceSourceEngine::startAcquisition() { call("cmdStart"); return m_state == StRunning; }
void GpuDeviceSourceEngine::stopAcquistion() { call("cmdStop"); }
void GpuDeviceSourceEngine::setSource(DeviceSampleSource* source) { call("cmdSetSource", source, true); }
void GpuDeviceSourceEngine::addSink(BasebandSampleSink* sink) { call("cmdAddSink", sink, true); }
void GpuDeviceSourceEngine::removeSink(BasebandSampleSink* sink) { call("cmdRemoveSink", sink, true); }
void GpuDeviceSourceEngine::addThreadedSink(ThreadedBasebandSampleSink* sink) { call("cmdAddThreadedSink", sink, true); }
void GpuDeviceSourceEngine::removeThreadedSink(ThreadedBasebandSampleSink* sink) { call("cmdRemoveThreadedSink", sink, true); }

void GpuDeviceSourceEngine::configureCorrections(bool dcOffsetCorrection, bool iqImbalanceCorrection)
{
    // asynchronous in the reference (a DSPConfigureCorrection pushed on the input queue, :152-157)
    QMetaObject::invokeMethod(this, "cmdConfigureCorrections", isRunning() && QThread::currentThread() != this ? Qt::QueuedConnection : Qt::DirectConnection,
                              Q_ARG(bool, dcOffsetCorrection), Q_ARG(bool, iqImbalanceCorrection));
}

QString GpuDeviceSourceEngine::errorMessage() { return m_errorMessage; }
QString GpuDeviceSourceEngine::sourceDeviceDescription() { return m_deviceDescription; }

// ---------------------------------------------------------------------------------------------- commands (engine thread)
void GpuDeviceSourceEngine::cmdInit()
{
    m_state = gotoIdle();                               // handleSynchronousMessages, DSPAcquisitionInit (:611-618)
    if (m_state == StIdle) m_state = gotoInit();
}

void GpuDeviceSourceEngine::cmdStart() { if (m_state == StReady) m_state = gotoRunning(); }
void GpuDeviceSourceEngine::cmdStop() { m_state = gotoIdle(); }

void GpuDeviceSourceEngine::cmdSetSource(void* p)
{
    gotoIdle();                                         // handleSetSource (:577-597)
    m_deviceSampleSource = static_cast<DeviceSampleSource*>(p);
    if (m_deviceSampleSource != 0)
        connect(m_deviceSampleSource->getSampleFifo(), SIGNAL(dataReady()), this, SLOT(handleData()), Qt::QueuedConnection);
}

void GpuDeviceSourceEngine::cmdAddSink(void* p)
{
    BasebandSampleSink* sink = static_cast<BasebandSampleSink*>(p);
    m_basebandSampleSinks.push_back(sink);
    DSPSignalNotification msg(m_sampleRate, m_centerFrequency);   // initialise rate / centre frequency in the sink (:649-655)
    sink->handleMessage(msg);
    if (m_state == StRunning) sink->start();
}

void GpuDeviceSourceEngine::cmdRemoveSink(void* p)
{
    BasebandSampleSink* sink = static_cast<BasebandSampleSink*>(p);
    if (m_state == StRunning) sink->stop();
    m_basebandSampleSinks.remove(sink);
}

void GpuDeviceSourceEngine::cmdAddThreadedSink(void* p)
{
    ThreadedBasebandSampleSink* sink = static_cast<ThreadedBasebandSampleSink*>(p);
    m_threadedBasebandSampleSinks.push_back(sink);
    DSPSignalNotification msg(m_sampleRate, m_centerFrequency);
    sink->handleSinkMessage(msg);
    if (m_state == StRunning) sink->start();
}

void GpuDeviceSourceEngine::cmdRemoveThreadedSink(void* p)
{
    ThreadedBasebandSampleSink* sink = static_cast<ThreadedBasebandSampleSink*>(p);
    sink->stop();
    m_threadedBasebandSampleSinks.remove(sink);
}

void GpuDeviceSourceEngine::cmdConfigureCorrections(bool dc, bool iq)
{
    // handleInputMessages, DSPConfigureCorrection (:694-725): the flags, and EVERY averaging member is reset
    m_iqImbalanceCorrection = iq;
    m_dcOffsetCorrection = dc;
    if (m_dccorr) sdrx_dccorr_reset(m_dccorr);
    if (m_iqimb) sdrx_iqimb_reset(m_iqimb);
}

// ---------------------------------------------------------------------------------------------- state machine
GpuDeviceSourceEngine::State GpuDeviceSourceEngine::gotoIdle()
{
    switch (m_state) {
    case StNotStarted: return StNotStarted;
    case StIdle: case StError: return StIdle;
    case StReady: case StRunning: break;
    }
    if (m_deviceSampleSource == 0) return StIdle;
    for (std::list<BasebandSampleSink*>::const_iterator it = m_basebandSampleSinks.begin(); it != m_basebandSampleSinks.end(); ++it) (*it)->stop();
    for (std::list<ThreadedBasebandSampleSink*>::const_iterator it = m_threadedBasebandSampleSinks.begin(); it != m_threadedBasebandSampleSinks.end(); ++it) (*it)->stop();
    m_deviceSampleSource->stop();
    m_deviceDescription.clear();
    m_sampleRate = 0;
    return StIdle;
}

GpuDeviceSourceEngine::State GpuDeviceSourceEngine::gotoInit()
{
    switch (m_state) {
    case StNotStarted: return StNotStarted;
    case StRunning: return StRunning;
    case StReady: return StReady;
    case StIdle: case StError: break;
    }
    if (m_deviceSampleSource == 0) return gotoError("No sample source configured");
    m_deviceDescription = m_deviceSampleSource->getDeviceDescription();
    m_centerFrequency = m_deviceSampleSource->getCenterFrequency();
    m_sampleRate = m_deviceSampleSource->getSampleRate();
    DSPSignalNotification notif(m_sampleRate, m_centerFrequency);
    for (std::list<BasebandSampleSink*>::const_iterator it = m_basebandSampleSinks.begin(); it != m_basebandSampleSinks.end(); ++it) (*it)->handleMessage(notif);
    for (std::list<ThreadedBasebandSampleSink*>::const_iterator it = m_threadedBasebandSampleSinks.begin(); it != m_threadedBasebandSampleSinks.end(); ++it) (*it)->handleSinkMessage(notif);
    if (m_deviceSampleSource->getMessageQueueToGUI())
        m_deviceSampleSource->getMessageQueueToGUI()->push(new DSPSignalNotification(notif));
    return StReady;
}

GpuDeviceSourceEngine::State GpuDeviceSourceEngine::gotoRunning()
{
    switch (m_state) {
    case StNotStarted: return StNotStarted;
    case StIdle: return StIdle;
    case StRunning: return StRunning;
    case StReady: case StError: break;
    }
    if (m_deviceSampleSource == 0) return gotoError("GpuDeviceSourceEngine::gotoRunning: No sample source configured");
    // the correction objects live on the GPU: no device, no engine (no CPU fallback)
    if (!m_dccorr && sdrx_dccorr_create(&m_dccorr, m_device) != SDRX_OK) return gotoError(QString("sdrx_dccorr_create: %1").arg(sdrx_last_error()));
    if (!m_iqimb && sdrx_iqimb_create(&m_iqimb, m_device, 1) != SDRX_OK) return gotoError(QString("sdrx_iqimb_create: %1").arg(sdrx_last_error()));
    if (!m_deviceSampleSource->start()) return gotoError("Could not start sample source");
    for (std::list<BasebandSampleSink*>::const_iterator it = m_basebandSampleSinks.begin(); it != m_basebandSampleSinks.end(); ++it) (*it)->start();
    for (std::list<ThreadedBasebandSampleSink*>::const_iterator it = m_threadedBasebandSampleSinks.begin(); it != m_threadedBasebandSampleSinks.end(); ++it) (*it)->start();
    return StRunning;
}

GpuDeviceSourceEngine::State GpuDeviceSourceEngine::gotoError(const QString& msg)
{
    m_errorMessage = msg;
    m_deviceDescription.clear();
    m_state = StError;
    return StError;
}

// ---------------------------------------------------------------------------------------------- data path
void GpuDeviceSourceEngine::handleData()
{
    if (m_state == StRunning) work();
}

void GpuDeviceSourceEngine::correct(SampleVector::iterator begin, SampleVector::iterator end)
{
    // iqCorrections(begin, end, m_iqImbalanceCorrection) (:175-262), in place on the FIFO span like the reference
    int16_t* iq = reinterpret_cast<int16_t*>(&*begin);
    const int64_t n = (int64_t)(end - begin);
    int rc;
    if (m_iqImbalanceCorrection) { int16_t* p[1] = { iq }; rc = sdrx_iqimb_process(m_iqimb, p, &n); }
    else rc = sdrx_dccorr_process(m_dccorr, iq, n);
    if (rc != SDRX_OK) gotoError(QString("GPU correction failed: %1").arg(sdrx_last_error()));
}

void GpuDeviceSourceEngine::work()
{
    SampleSinkFifo* sampleFifo = m_deviceSampleSource->getSampleFifo();
    std::size_t samplesDone = 0;
    const bool positiveOnly = false;
    while ((sampleFifo->fill() > 0) && (samplesDone < m_sampleRate) && (m_state == StRunning))
    {
        SampleVector::iterator part[4];
        const std::size_t count = sampleFifo->readBegin(sampleFifo->fill(), &part[0], &part[1], &part[2], &part[3]);
        for (int p = 0; p < 4; p += 2) {                 // first part, then the wrapped-around part
            if (part[p] == part[p + 1]) continue;
            if (m_dcOffsetCorrection) correct(part[p], part[p + 1]);
            if (m_state != StRunning) break;             // a GPU error took the engine to StError
            for (std::list<BasebandSampleSink*>::const_iterator it = m_basebandSampleSinks.begin(); it != m_basebandSampleSinks.end(); ++it)
                (*it)->feed(part[p], part[p + 1], positiveOnly);
            for (std::list<ThreadedBasebandSampleSink*>::const_iterator it = m_threadedBasebandSampleSinks.begin(); it != m_threadedBasebandSampleSinks.end(); ++it)
                (*it)->feed(part[p], part[p + 1], positiveOnly);
        }
        sampleFifo->readCommit((unsigned int) count);
        samplesDone += count;
    }
}
