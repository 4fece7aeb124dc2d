#include "resblock_kernel_h2.h"
namespace ttsamd {
int resblock_pair_launch_k3(const ttsamd_resblock_args &a, hipStream_t st) { return resblock_pair_launch_k<3>(a, st); }
}  // namespace ttsamd
#ifdef TTSAMD_PHASE_CLOCKS
TTSAMD_RES_CLOCK_GETTER(ttsamd_debug_res_clocks_k3)
#endif
