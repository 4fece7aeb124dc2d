#include "conv_dispatch.h"
namespace ttsamd {
int conv1d_launch_k1(const ttsamd_conv1d_args &a, hipStream_t st) { return conv1d_launch_kd<1, 1>(a, st); }
}  // namespace ttsamd
