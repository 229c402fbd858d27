// Shadow of sdrbase/dsp/decimators.h for building reference sources UNCHANGED against the GPU classes: put
// `-I<sdrx>/qt_adapter/shadow` in front of `-I<reference>/sdrbase` and `Decimators<...>` becomes sdrx::Decimators<...>
// (include/sdrx/dsp.hpp) on the reference's own Sample / SampleVector (dsp/dsptypes.h is NOT shadowed).
#ifndef SDRX_SHADOW_DECIMATORS_H
#define SDRX_SHADOW_DECIMATORS_H
#include "dsp/dsptypes.h"
#ifndef SDRX_HOST_SAMPLE
#define SDRX_HOST_SAMPLE ::Sample
#endif
#ifndef SDRX_HOST_FSAMPLE
#define SDRX_HOST_FSAMPLE ::FSample
#endif
#include "sdrx/dsp.hpp"
template<typename StorageType, typename T, uint SdrBits, uint InputBits>
using Decimators = sdrx::Decimators<StorageType, T, SdrBits, InputBits>;
#endif
