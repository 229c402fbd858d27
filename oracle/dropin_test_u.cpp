// TEST INFRASTRUCTURE ONLY.  Drop-in harness, DecimatorsU part (own translation unit: see dropin_common.hpp).
#include <stdint.h>
#include <vector>
#include "dsp/dsptypes.h"
#include "dsp/decimatorsu.h"
#define SDRX_HOST_SAMPLE ::Sample
#include "sdrx/dsp.hpp"
#include "dropin_common.hpp"

void producer_side_u(int device)
{
    typedef DecimatorsU<qint32, quint8, SDR_RX_SAMP_SZ, 8, 127> RefDecU;
    typedef sdrx::DecimatorsU<qint32, quint8, SDR_RX_SAMP_SZ, 8, 127> GpuDecU;
    PRODUCER(RefDecU, GpuDecU, quint8, decimate16_sup, 0, 256)              // RTL-SDR thread (rtlsdrthread.h:55)
    PRODUCER(RefDecU, GpuDecU, quint8, decimate64_cen, 0, 256)
    PRODUCER(RefDecU, GpuDecU, quint8, decimate2_inf, 0, 256)
    PRODUCER(RefDecU, GpuDecU, quint8, decimate1, 0, 256)
}
