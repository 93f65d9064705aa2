// kernels + launchers for groups of 4 lanes x 1 mask words per lane
#define TNCO_INST_L 2
#define TNCO_INST_K 1
#include "launch_impl.h"
