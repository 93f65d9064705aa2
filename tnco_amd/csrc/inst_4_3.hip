// kernels + launchers for groups of 16 lanes x 3 mask words per lane
#define TNCO_INST_L 4
#define TNCO_INST_K 3
#include "launch_impl.h"
