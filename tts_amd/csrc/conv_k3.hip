#include "conv_dispatch.h"
namespace ttsamd {
int conv1d_launch_k3(const ttsamd_conv1d_args &a, hipStream_t st)
{
    switch (a.dilation) {
        case 1: return conv1d_launch_kd<3, 1>(a, st);
        case 3: return conv1d_launch_kd<3, 3>(a, st);
        case 5: return conv1d_launch_kd<3, 5>(a, st);
        case 9: return conv1d_launch_kd<3, 9>(a, st);
    }
    return TTSAMD_ERR_UNSUPPORTED;
}
}  // namespace ttsamd
