// kernels + launchers for groups of 8 lanes x 3 mask words per lane
#define TNCO_INST_L 3
#define TNCO_INST_K 3
#include "launch_impl.h"
