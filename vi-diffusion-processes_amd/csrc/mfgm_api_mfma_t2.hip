// 2 x 2 tiles (16 < d <= 32)
#define MFGM_MFMA_NT 2
#include "mfgm_mfma_launch.h"
