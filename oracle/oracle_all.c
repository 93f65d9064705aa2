/*
 * oracle/oracle_all.c -- TEST INFRASTRUCTURE ONLY (see tnco_oracle.c header).
 * Instantiates the CPU restatement for cost_type float64 and float32
 * (include/tnco/globals.hpp:81-84 EXPAND_COST_TYPE; long double / float1024
 * are not restated).
 */
#define COST_T double
#define PFX(name) orc_f64_##name
#include "tnco_oracle.c"
#undef COST_T
#undef PFX
#define COST_T float
#define PFX(name) orc_f32_##name
#include "tnco_oracle.c"
