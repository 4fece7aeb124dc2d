// Log-likelihood matrix that feeds the monotonic alignment search (Vits.forward_mas, vits.py:912-918;
// GlowTTS.forward / inference_with_MAS, glow_tts.py:241-247,291-296):
//   logp[b,tx,ty] = sum_c exp(-2 logs[c,tx]) * (-0.5 z[c,ty]^2)            (logp2)
//                 + sum_c (m[c,tx] exp(-2 logs[c,tx])) * z[c,ty]            (logp3)
//                 + sum_c (-0.5 log(2 pi) - logs[c,tx])                     (logp1)
//                 + sum_c (-0.5 m[c,tx]^2 exp(-2 logs[c,tx]))               (logp4)
// Two batched K=C contractions: fp32-input MFMA 32x32x2 (exact fp32 products).  Both operands are contiguous
// along their non-contracted axis in the channels-first layout (tx for the text side, ty for the latent), i.e.
// already in MFMA fragment lane order: fragments are built on the fly from coalesced global loads (the exp /
// square live in the load path), no LDS staging, output rows stored coalesced along ty.  Replaces two einsums /
// matmuls + ~8 elementwise torch ops and keeps `logp` on the device for ttsamd_maximum_path (the reference moves it
// to the CPU, helpers.py:187).
#include "common.h"

namespace ttsamd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(256) void mas_logp_kernel(float *__restrict__ logp, const float *__restrict__ z,
                                                       const float *__restrict__ m, const float *__restrict__ logs,
                                                       int C, int Tx, int Ty, int glow_order)
{
    __shared__ float red[2][8][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, j = lane & 31;
    const int b = blockIdx.z;
    const int tx0 = blockIdx.y * 32;
    const int ty = (blockIdx.x * 4 + wave) * 32 + j;
    const float *mb = m + (long)b * C * Tx, *lb = logs + (long)b * C * Tx, *zb = z + (long)b * C * Ty;

    // per-row scalars logp1 / logp4 (8 partial sums per row, fixed-order reduction)
    {
        const int i = tid & 31, part = tid >> 5;
        const int tx = tx0 + i;
        float p1 = 0.f, p4 = 0.f;
        if (tx < Tx)
            for (int c = part; c < C; c += 8) {
                const float l = lb[(long)c * Tx + tx], mm = mb[(long)c * Tx + tx];
                p1 += -0.91893853320467274178f - l;            // -0.5*log(2*pi) - logs
                p4 += -0.5f * (mm * mm) * expf(-2.f * l);
            }
        red[0][part][i] = p1;
        red[1][part][i] = p4;
    }
    __syncthreads();

    f32x16 acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc2[r] = 0.f; acc3[r] = 0.f; }
    const int txa = tx0 + j;
    const bool av = txa < Tx, bv = ty < Ty;
    for (int c2 = 0; c2 < C; c2 += 2) {
        const int c = c2 + hh;
        const bool cv = c < C;
        const float l = (av && cv) ? lb[(long)c * Tx + txa] : 0.f;
        const float mm = (av && cv) ? mb[(long)c * Tx + txa] : 0.f;
        const float zz = (bv && cv) ? zb[(long)c * Ty + ty] : 0.f;
        const float os = (av && cv) ? expf(-2.f * l) : 0.f;
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(os, -0.5f * (zz * zz), acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(mm * os, zz, acc3, 0, 0, 0);
    }
    if (bv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (tx0 + i < Tx) {
                float l1 = red[0][0][i], l4 = red[1][0][i];
#pragma unroll
                for (int p = 1; p < 8; ++p) { l1 += red[0][p][i]; l4 += red[1][p][i]; }
                const float v = glow_order ? (((l1 + acc2[r]) + acc3[r]) + l4)     // logp1 + logp2 + logp3 + logp4
                                           : (((acc2[r] + acc3[r]) + l1) + l4);     // logp2 + logp3 + logp1 + logp4
                logp[((long)b * Tx + tx0 + i) * Ty + ty] = v;
            }
        }
    }
}

}  // namespace ttsamd
using namespace ttsamd;

extern "C" int ttsamd_mas_logp(float *logp, const float *z, const float *m, const float *logs, int batch, int c,
                               int t_x, int t_y, int glow_order, void *stream)
{
    TTSAMD_CHECK_ARG(logp && z && m && logs && batch >= 0 && c > 0 && t_x >= 0 && t_y >= 0, "mas_logp: bad args");
    if (batch == 0 || t_x == 0 || t_y == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535 && (t_x + 31) / 32 <= 65535, "mas_logp: shape too large");
    hipLaunchKernelGGL(mas_logp_kernel, dim3((t_y + 127) / 128, (t_x + 31) / 32, batch), dim3(256), 0, as_stream(stream),
                       logp, z, m, logs, c, t_x, t_y, glow_order);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
