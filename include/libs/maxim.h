// include/libs/maxim.h -- the reference's umbrella header for its analysis classes (src/libs/maxim.h): patches that include it
// next to "maximilian.h" (e.g. maximilian_examples/20.FFT_example) find maxiFFT / maxiIFFT / maxiMFCC in the drop-in header.
#pragma once
#include "../maximilian.h"
