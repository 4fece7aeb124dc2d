// Single-output-channel Conv1d (HiFiGAN conv_post: C -> 1, k = 7, leaky-ReLU in, tanh out;
// TTS/vocoder/models/hifigan_generator.py:262-264) as a streaming kernel.
//
// With ONE output row the implicit-GEMM kernels spend a 32-row MFMA tile on it and the layer runs at 1.5 TB/s; it is pure
// HBM streaming (C*4 bytes read + 4 written per sample, 3.4 FLOP/B), so it is written as such: a lane owns 4 consecutive
// columns, reads each input row with one 16-byte load, takes the three neighbour columns either side from the adjacent
// lanes with DPP row/wave shifts (no LDS, no re-reads), and runs the 7-tap FMA chain in exact fp32.  A wave covers 256
// loaded columns of which the inner 248 are outputs (the two edge lanes only feed their neighbours).  Weights are read
// from the packed fp32 image with scalar loads.
#include "conv_kernel.h"

namespace ttsamd {

using f32x4u = __attribute__((ext_vector_type(4), aligned(4))) float;   // 16-byte vector at 4-byte alignment

constexpr int kPostThreads = 256;
constexpr int kPostColsPerWave = 248;   // outputs per wave: lanes 1..62 x 4 columns
constexpr int kPostWavesPerBlock = kPostThreads / 64;

__device__ __forceinline__ float post_shr(float v)   // value of lane - 1 (0 into lane 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}

// index of W[0, ci, tap] inside the packed image of ttsamd_conv1d_pack_weights (c_out <= 32: one m-tile)
__device__ __forceinline__ int post_widx(int ci, int tap, int K)
{
    const int c = ci / kConvCK, p = (ci % kConvCK) / 2;
    const int ksl = p * K + tap;
    const int g = c * ((kConvCK / 2) * K / 4) + ksl / 4;
    return (g * 64 + (ci & 1) * 32) * 4 + (ksl & 3);
}

template <int K, int CU>
__global__ __launch_bounds__(kPostThreads) void conv_post_kernel(const ttsamd_conv1d_args a)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    const int w0 = (blockIdx.x * kPostWavesPerBlock + wave) * kPostColsPerWave;   // first output column of this wave
    if (w0 >= a.t_out) return;
    const int t = w0 - 5 + 4 * lane;                                              // first of this lane's 4 INPUT columns
    static_assert(K == 7, "the neighbour exchange below is written for 7 taps (6 columns to the left)");
    const float *xb = a.x + (long)b * a.x_bstride;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, ((long)(a.c_in - 1) * a.x_rstride + a.t_in) * 4);
    // columns outside [0, t_in) read as 0 (the conv's zero padding); a 16-byte load straddling the end is done as dwords
    const bool whole = (t >= 0) && (t + 3 < a.t_in);
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.in_mask) {
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (t + i >= 0 && t + i < a.t_in) ? a.in_mask[(long)b * a.t_in + t + i] : 0.f;
    }
    const float slope = a.in_slope;
    const bool act = a.in_act == TTSAMD_ACT_LRELU;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int rs4 = (int)a.x_rstride * 4;
#pragma unroll 1
    for (int c0 = 0; c0 < a.c_in; c0 += CU) {
        float v[CU][4];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int c = c0 + u;
            const int off = (c < a.c_in) ? c * rs4 + t * 4 : kConvOob;
            if (whole && c < a.c_in) {
                // 16 bytes at a 4-byte-aligned address: a GLOBAL load (a 16-byte BUFFER load at such an offset returns
                // wrong data on gfx950 — measured; global loads run in unaligned-access mode)
                const f32x4u q = *reinterpret_cast<const f32x4u *>(xb + (long)c * a.x_rstride + t);
                v[u][0] = q[0];
                v[u][1] = q[1];
                v[u][2] = q[2];
                v[u][3] = q[3];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[u][i] = ld_buf(rx, (c < a.c_in && t + i >= 0 && t + i < a.t_in) ? off + 4 * i : kConvOob, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int c = c0 + u;
            if (c < a.c_in) {
                float x[10];    // input columns t-6 .. t+3
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float s = v[u][i] * m[i];
                    x[6 + i] = act ? (s > 0.f ? s : s * slope) : s;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) x[2 + i] = post_shr(x[6 + i]);   // lane-1's four columns: t-4 .. t-1
                x[0] = post_shr(x[4]);                                       // lane-2's columns 2, 3: t-6, t-5
                x[1] = post_shr(x[5]);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float w = a.w_packed[post_widx(c, k, K)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(w, x[i + k], acc[i]);
                }
            }
        }
    }
    if (lane >= 2) {                       // output column of acc[i]: t - 3 + i (its window is input columns t-6+i .. t+i)
        const float bias = a.bias ? a.bias[0] : 0.f;
        float *y = a.y + (long)b * a.y_bstride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tt = t - 3 + i;
            if (tt < a.t_out) {
                float o = acc[i] + bias;
                if (a.out_act == TTSAMD_ACT_TANH) o = tanhf(o);
                else if (a.out_act == TTSAMD_ACT_RELU) o = fmaxf(o, 0.f);
                y[tt] = o;
            }
        }
    }
}

bool conv_post_eligible(const ttsamd_conv1d_args &a)
{
    return a.mode == TTSAMD_CONV_NORMAL && a.c_out == 1 && a.kernel == 7 && a.dilation == 1 && a.pad_left == 3 &&
           a.t_in == a.t_out && !a.res && !a.accum && !a.out_mask && !a.row_bias && a.out_div == 0.f && a.c_in <= 128;
}

int conv_post_launch(const ttsamd_conv1d_args &a, hipStream_t st)
{
    const int waves = (a.t_out + kPostColsPerWave - 1) / kPostColsPerWave;
    const int blocks = (waves + kPostWavesPerBlock - 1) / kPostWavesPerBlock;
    hipLaunchKernelGGL((conv_post_kernel<7, 8>), dim3(blocks, a.batch), dim3(kPostThreads), 0, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

}  // namespace ttsamd
