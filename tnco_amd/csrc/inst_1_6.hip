// kernels + launchers for groups of 2 lanes x 6 mask words per lane
#define TNCO_INST_L 1
#define TNCO_INST_K 6
#include "launch_impl.h"
