// Shadow of sdrbase/dsp/decimatorsu.h (see decimators.h next to this file): the RTL-SDR thread's
// DecimatorsU<qint32, quint8, SDR_RX_SAMP_SZ, 8, 127> (plugins/samplesource/rtlsdr/rtlsdrthread.h:55) becomes sdrx::DecimatorsU.
#ifndef SDRX_SHADOW_DECIMATORSU_H
#define SDRX_SHADOW_DECIMATORSU_H
#include "dsp/decimators.h"
template<typename StorageType, typename T, uint SdrBits, uint InputBits, int Shift>
using DecimatorsU = sdrx::DecimatorsU<StorageType, T, SdrBits, InputBits, Shift>;
#endif
