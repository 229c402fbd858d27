// Qt adapter: a device-set engine with the public face of DSPDeviceSourceEngine (sdrbase/dsp/dspdevicesourceengine.h:48-77)
// whose work() hands the device FIFO spans to the GPU library:
//
//     readBegin -> [sdrx_dccorr / sdrx_iqimb on the span, in place] -> direct sinks (a GpuDownChannelizerBank is one of
//     them: ONE feed per span for all its channels; spectrum, FileRecord ...) -> threaded sinks -> readCommit
//
// Same thread model as the reference: the engine is a QThread that owns its event loop (moveToThread(this)); commands
// from other threads execute IN the engine thread and the caller blocks until they are done (the reference does this
// with SyncMessenger::sendWait, here a blocking queued invocation); FIFO data arrives through the queued dataReady()
// connection; configureCorrections() is asynchronous like the reference's message-queue post.  Same state machine
// (notStarted -> idle -> ready -> running, error), same notifications to the sinks (DSPSignalNotification at init and
// on addSink).  A GPU failure takes the engine to StError with the library's error text (gotoError, :567-575).
#ifndef SDRX_QT_GPUDEVICESOURCEENGINE_H
#define SDRX_QT_GPUDEVICESOURCEENGINE_H

#include <QThread>
#include <QString>
#include <list>
#include "dsp/dsptypes.h"
#include "sdrx.h"

class DeviceSampleSource;
class BasebandSampleSink;
class ThreadedBasebandSampleSink;

class GpuDeviceSourceEngine : public QThread {
    Q_OBJECT
public:
    enum State { StNotStarted, StIdle, StReady, StRunning, StError };     // DSPDeviceSourceEngine::State

    explicit GpuDeviceSourceEngine(uint uid, int device = 0, QObject* parent = NULL);
    ~GpuDeviceSourceEngine();

    uint getUID() const { return m_uid; }

    void start();                       //!< this thread start
    void stop();                        //!< this thread stop

    bool initAcquisition();             //!< idle -> ready: DSPSignalNotification(rate, fc) to every sink
    bool startAcquisition();            //!< ready -> running: source and sinks started
    void stopAcquistion();              //!< (sic) -> idle

    void setSource(DeviceSampleSource* source);
    DeviceSampleSource* getSource() { return m_deviceSampleSource; }

    void addSink(BasebandSampleSink* sink);
    void removeSink(BasebandSampleSink* sink);
    void addThreadedSink(ThreadedBasebandSampleSink* sink);
    void removeThreadedSink(ThreadedBasebandSampleSink* sink);

    void configureCorrections(bool dcOffsetCorrection, bool iqImbalanceCorrection);

    State state() const { return m_state; }
    QString errorMessage();
    QString sourceDeviceDescription();

private slots:
    void handleData();                                                    //!< dataReady() of the source FIFO
    void cmdInit(); void cmdStart(); void cmdStop();
    void cmdSetSource(void* source);
    void cmdAddSink(void* sink); void cmdRemoveSink(void* sink);
    void cmdAddThreadedSink(void* sink); void cmdRemoveThreadedSink(void* sink);
    void cmdConfigureCorrections(bool dc, bool iq);

private:
    void run();
    void work();
    void correct(SampleVector::iterator begin, SampleVector::iterator end);
    State gotoIdle(); State gotoInit(); State gotoRunning(); State gotoError(const QString& msg);
    void call(const char* slot, void* arg = 0, bool hasArg = false);     //!< run a command slot in the engine thread, wait for it

    uint m_uid;
    int m_device;
    State m_state;
    QString m_errorMessage, m_deviceDescription;
    DeviceSampleSource* m_deviceSampleSource;
    std::list<BasebandSampleSink*> m_basebandSampleSinks;
    std::list<ThreadedBasebandSampleSink*> m_threadedBasebandSampleSinks;
    uint m_sampleRate;
    quint64 m_centerFrequency;
    bool m_dcOffsetCorrection, m_iqImbalanceCorrection;
    sdrx_dccorr_t* m_dccorr;            //!< DC only      (iqCorrections(.., false))
    sdrx_iqimb_t* m_iqimb;              //!< DC + I/Q imbalance (iqCorrections(.., true)), one stream
};

#endif
