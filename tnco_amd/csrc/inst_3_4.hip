// kernels + launchers for groups of 8 lanes x 4 mask words per lane
#define TNCO_INST_L 3
#define TNCO_INST_K 4
#include "launch_impl.h"
