// Shadow of sdrbase/dsp/decimatorsfi.h (see decimators.h next to this file)
#ifndef SDRX_SHADOW_DECIMATORSFI_H
#define SDRX_SHADOW_DECIMATORSFI_H
#include "dsp/decimators.h"
typedef sdrx::DecimatorsFI DecimatorsFI;
#endif
