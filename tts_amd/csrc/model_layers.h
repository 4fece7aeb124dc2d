// Layer drivers shared by the acoustic-model handles (vits_model.hip, glow_model.hip): the C++ counterparts of tts_amd/layers.py —
// weight preparation at finalize and fixed launch sequences over the kernel-level ABI.  Host code only.
#pragma once
#include <algorithm>

#include "model_common.h"

namespace ttsamd {
namespace model {

constexpr int kPairRows = 16;            // paired-row conv modes: a packed 32-row tile = 16 first halves + the 16 matching second halves

struct Norm {
    DevBuf gamma, beta;
    float eps = 1e-5f;
};
int upload_norm(const TensorMap &t, const char *who, const std::string &name, int c, float eps, Norm &n);

// rows of w [rows, inner] re-ordered for the paired-row modes (tts_amd/ops.py: pair_index / pair_permute): tile a holds rows
// [16a, 16a + 16) of the first operand followed by the same channels of the second (+ second_offset); zero rows pad a partial tile
void pair_permute(const std::vector<float> &w, const float *bias, int64_t inner, int n, int second_offset, std::vector<float> &wp, std::vector<float> &bp);

// launch context of one request
struct Ctx {
    int precision;       // 0 h2, 1 x3, 2 f32
    void *s;             // stream
    int B, T;
};
inline int conv(const Ctx &c, ttsamd_conv1d_args &a) { return ttsamd_conv1d(&a, c.s); }
// y = [post_res +] act(LN_c([dwconv](x))) [* out_mask] on [B, ch, t] (tts_amd/ops.py: channel_norm)
int norm(const Ctx &c, const float *x, float *y, int ch, int t, const Norm &n, int act = TTSAMD_ACT_NONE, const float *out_mask = nullptr,
         const float *post_res = nullptr, const float *dw_w = nullptr, const float *dw_b = nullptr, int dw_kernel = 0, int dw_dilation = 1,
         const float *in_mask = nullptr);

// WaveNet block (TTS/tts/layers/generic/wavenet.py:16-123; tts_amd/layers.py: WN), no speaker conditioning
struct Wn {
    std::vector<std::unique_ptr<PackedConv>> in_layers, rs_layers;
};
int build_wn(const TensorMap &t, const char *who, const std::string &p, int hidden, int kernel, int dilation_rate, int layers, Wn &wn);
// h [B,H,t] is updated IN PLACE layer by layer; `out` receives the sum of skips * mask; the tanh * sigmoid gate (wavenet.py:6-13)
// lives in the in_layer conv's epilogue
int run_wn(const Ctx &c, const Wn &w, float *h, float *acts, float *out, const float *mask, int H, int t);

// relative-position transformer (TTS/tts/layers/glow_tts/transformer.py:322-432; tts_amd/layers.py: RelativePositionTransformer)
struct EncLayer {
    PackedConv qkv, o, f1, f2;
    DevBuf emb_k, emb_v;
    Norm n1, n2;
};
struct Transformer {
    std::vector<std::unique_ptr<EncLayer>> layers;
    int hidden = 0, ffn = 0, heads = 0, window = 0;       // window 0: rel_attn_window_size None (plain attention)
};
int build_transformer(const TensorMap &t, const char *who, const std::string &p, int hidden, int ffn, int heads, int layers, int kernel, int window,
                      float eps, Transformer &out);
struct TransformerBufs {
    float *xa, *xb, *qkv, *att, *xy, *x1, *hid, *y;
    void take(Bump &ws, size_t n, int H, int F)
    {
        xa = ws.take(n * H), xb = ws.take(n * H), qkv = ws.take(n * 3 * H), att = ws.take(n * H), xy = ws.take(n * H), x1 = ws.take(n * H);
        hid = ws.take(n * F), y = ws.take(n * H);
    }
};
// x (in bufs.xa, already multiplied by mask) -> the buffer holding the result (xa or xb by layer parity)
int run_transformer(const Ctx &c, const Transformer &tr, const TransformerBufs &b, const float *mask, float **out);
inline float *transformer_result(const Transformer &tr, const TransformerBufs &b) { return tr.layers.size() % 2 == 0 ? b.xa : b.xb; }

// DurationPredictor (TTS/tts/layers/glow_tts/duration_predictor.py:7-69; tts_amd/layers.py: DurationPredictor), no conditioning
struct Dp {
    PackedConv c1, c2, proj;
    Norm n1, n2;
    int hidden = 256;
};
int build_dp(const TensorMap &t, const char *who, const std::string &p, int in_channels, int hidden, Dp &dp);
// x [B,C,T] -> logw [B,T] (= the [B,1,T] projection); d[0..3]: four [B,hidden,T] buffers
int run_dp(const Ctx &c, const Dp &dp, const float *x, int c_x, const float *mask, float *const d[4], float *logw);

}  // namespace model
}  // namespace ttsamd
