#include "resblock_kernel_x3.h"
namespace ttsamd {
int resblock_group_launch_d5(const ResGroupArgs &g, int c, int t, int batch, hipStream_t st) { return resblock_group_launch_d<5>(g, c, t, batch, st); }
}  // namespace ttsamd
