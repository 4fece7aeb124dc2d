#include "conv_dispatch.h"
namespace ttsamd {
int conv1d_launch_k7(const ttsamd_conv1d_args &a, hipStream_t st)
{
    switch (a.dilation) {
        case 1: return conv1d_launch_kd<7, 1>(a, st);
        case 3: return conv1d_launch_kd<7, 3>(a, st);
        case 5: return conv1d_launch_kd<7, 5>(a, st);
    }
    return TTSAMD_ERR_UNSUPPORTED;
}
}  // namespace ttsamd
