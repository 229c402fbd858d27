// Shadow of sdrbase/dsp/decimatorsif.h (see decimators.h next to this file)
#ifndef SDRX_SHADOW_DECIMATORSIF_H
#define SDRX_SHADOW_DECIMATORSIF_H
#include "dsp/decimators.h"
template<typename T, uint InputBits> using DecimatorsIF = sdrx::DecimatorsIF<T, InputBits>;
#endif
