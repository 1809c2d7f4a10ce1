// flop-counting build of the oracle (make -C oracle count): tmjx_oracle.c / tmjx_oracle_env.c are compiled as C++ with
// -DORACLE_COUNT, i.e. `real` = the counting class of counted_real.hpp; this file owns the counter and its C accessors
#include "counted_real.hpp"
thread_local OCount g_ocount = {0, 0, 0, 0, 0};
extern "C" void oracle_count_get(unsigned long long *out) { out[0] = g_ocount.add; out[1] = g_ocount.mul; out[2] = g_ocount.div; out[3] = g_ocount.special; out[4] = g_ocount.minmax; }
extern "C" void oracle_count_reset(void) { g_ocount = OCount{0, 0, 0, 0, 0}; }
