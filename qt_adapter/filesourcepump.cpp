#include "filesourcepump.h"
#include <cstring>

FileSourcePump::FileSourcePump(const QString& fileName) :
    m_fileName(fileName), m_deviceDescription("FileSource (sdrx pump)"), m_samplesCount(0), m_running(false)
{
    std::memset(&m_header, 0, sizeof m_header);
}

FileSourcePump::~FileSourcePump() { stop(); }

bool FileSourcePump::readHeader()
{
    std::ifstream f(m_fileName.toStdString().c_str(), std::ios::binary);
    quint8 raw[SDRX_SDRIQ_HEADER_BYTES];
    if (!f.read(reinterpret_cast<char*>(raw), sizeof raw)) return false;
    return sdrx_sdriq_parse_header(raw, sizeof raw, &m_header) == SDRX_OK;            // FileRecord::readHeader (filerecord.cpp:141-148)
}

bool FileSourcePump::start()
{
    if (!readHeader()) return false;
    m_ifstream.open(m_fileName.toStdString().c_str(), std::ios::binary);
    if (!m_ifstream.is_open()) return false;
    m_ifstream.seekg(SDRX_SDRIQ_HEADER_BYTES, std::ios::beg);                          // the stream continues behind the 24 header bytes
    m_sampleFifo.setSize(m_header.sample_rate * 4 > 0 ? (quint32) m_header.sample_rate * 4 : 1 << 20);   // filesourceinput.cpp:143
    m_samplesCount = 0;
    m_running = true;
    return true;
}

void FileSourcePump::stop()
{
    m_running = false;
    if (m_ifstream.is_open()) m_ifstream.close();
}

unsigned int FileSourcePump::tick(int throttleMs)
{
    if (!m_running) return 0;
    const std::size_t sampleBytes = m_header.sample_size > 16 ? sizeof(int32_t) : sizeof(int16_t);
    const std::size_t chunk = 2 * sampleBytes * (((std::size_t) m_header.sample_rate * (std::size_t) throttleMs) / 1000);   // filesourcethread.cpp:183
    if (chunk > m_fileBuf.size()) m_fileBuf.resize(chunk);
    m_ifstream.read(reinterpret_cast<char*>(&m_fileBuf[0]), (std::streamsize) chunk);
    std::size_t nbBytes = chunk;
    if (m_ifstream.eof()) {
        nbBytes = (std::size_t) m_ifstream.gcount();
        m_ifstream.clear();
        m_ifstream.seekg(SDRX_SDRIQ_LOOP_OFFSET, std::ios::beg);                       // sizeof(FileRecord::Header) == 32 (:197)
        m_samplesCount = 0;
    } else {
        m_samplesCount += chunk / (2 * sampleBytes);
    }
    // writeToSampleFifo (:212-251), 16-bit build (SDR_RX_SAMP_SZ == 16)
    if (m_header.sample_size == 16) return m_sampleFifo.write(&m_fileBuf[0], (uint) nbBytes);
    const std::size_t nbSamples = nbBytes / (2 * sampleBytes);
    if (nbSamples * sizeof(Sample) > m_convertBuf.size()) m_convertBuf.resize(nbSamples * sizeof(Sample));
    FixReal* conv = reinterpret_cast<FixReal*>(&m_convertBuf[0]);
    const int32_t* fb = reinterpret_cast<const int32_t*>(&m_fileBuf[0]);
    for (std::size_t is = 0; is < nbSamples; is++) { conv[2 * is] = fb[2 * is] >> 8; conv[2 * is + 1] = fb[2 * is + 1] >> 8; }
    return m_sampleFifo.write(&m_convertBuf[0], (uint)(nbSamples * sizeof(Sample)));
}
