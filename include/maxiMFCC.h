// include/maxiMFCC.h -- patches written against the reference include "maxiMFCC.h" (src/libs/maxiMFCC.h) next to "maximilian.h":
// maxiMFCC lives in the drop-in header.
#pragma once
#include "maximilian.h"
