// include/maxiFFT.h -- patches written against the reference include "maxiFFT.h" (src/libs/maxiFFT.h) next to "maximilian.h":
// maxiFFT / maxiIFFT live in the drop-in header.
#pragma once
#include "maximilian.h"
