// 1 x 1 tiles (8 < d <= 16)
#define MFGM_MFMA_NT 1
#include "mfgm_mfma_launch.h"
