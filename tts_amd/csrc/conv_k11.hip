#include "conv_dispatch.h"
namespace ttsamd {
int conv1d_launch_k11(const ttsamd_conv1d_args &a, hipStream_t st)
{
    switch (a.dilation) {
        case 1: return conv1d_launch_kd<11, 1>(a, st);
        case 3: return conv1d_launch_kd<11, 3>(a, st);
        case 5: return conv1d_launch_kd<11, 5>(a, st);
    }
    return TTSAMD_ERR_UNSUPPORTED;
}
}  // namespace ttsamd
