// Shadow of sdrbase/dsp/decimatorsff.h (see decimators.h next to this file)
#ifndef SDRX_SHADOW_DECIMATORSFF_H
#define SDRX_SHADOW_DECIMATORSFF_H
#include "dsp/decimators.h"
typedef sdrx::DecimatorsFF DecimatorsFF;
#endif
