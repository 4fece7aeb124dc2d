#include "conv_dispatch.h"
namespace ttsamd {
int conv1d_launch_k2(const ttsamd_conv1d_args &a, hipStream_t st) { return conv1d_launch_kd<2, 1>(a, st); }
}  // namespace ttsamd
