// Model-level C ABI of the vocoder ("mi355_hifigan_{create,load,forward,destroy}"): a HifiganGenerator behind ONE
// handle — weight-norm folding, polyphase re-ordering of the transposed convs, packing into the three fragment images, the
// length masks, the launch sequence of TTS/vocoder/models/hifigan_generator.py:236-282 and (optionally) its replay as a hipGraph —
// so that a host that is not Python can run the model, and so that a request pays one C call instead of ~100 ctypes
// marshalling calls.  Every launch goes through the kernel-level ABI of this same library (ttsamd_conv1d, ttsamd_resblock_pair,
// ttsamd_resblock_group, ttsamd_sum_div, ttsamd_stage_masks, ttsamd_replicate_pad*): the arithmetic, the tile choices and therefore
// the bits are those of the Python-driven path (tts_amd/hifigan.py), which tests/test_hifigan_gpu.py checks.
//
// Ownership: the caller owns mel / lengths / wav (device pointers); the handle owns its packed weights (device) and a grow-only
// activation workspace (device).  One caller and one stream at a time per handle (the reference's Synthesizer is not thread-safe
// either, synthesizer.py:302-304); errors are return codes + ttsamd_last_error(), nothing throws across the ABI.
#include <cstdio>
#include <cstdlib>

#include "model_common.h"

using namespace ttsamd;
using namespace ttsamd::model;

namespace {

constexpr float kLreluSlope = 0.1f;       // hifigan_generator.py:11

struct VocGraph {
    const void *mel, *lengths, *in_mask;
    void *wav;
    int batch, frames;
    hipStream_t stream;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct Model {
    ttsamd_hifigan_config cfg{};
    std::map<std::string, HostTensor> tensors;
    std::map<std::string, std::unique_ptr<PackedConv>> convs;
    bool finalized = false;
    int hop = 1;
    DevBuf work;                 // activation workspace, grow-only
    std::vector<VocGraph> graphs;
    hipStream_t cap_stream = nullptr;   // the sequence is RECORDED on this stream (the caller's may be the NULL stream, which cannot
                                        // capture) and replayed on the caller's
    int precision = 0;           // 0 = h2 (three fp16 products on large grids), 1 = x3 (six bf16 products), 2 = f32
    // MRF branch streams: the resblocks of a stage are independent until their last conv, which chains the accumulate r1 + r2 + r3 in
    // the reference's order (hifigan_generator.py:255-261): each branch runs on its own HIP stream so that one branch's launch tail /
    // ramp overlaps another's compute, events order only the accumulating convs (tts_amd/hifigan.py: forward).  For a lone request;
    // a host with several requests in flight (request lanes) turns it off (TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES).
    bool concurrent = true;
    hipStream_t side[TTSAMD_HIFIGAN_MAX_KERNELS] = {};
    std::vector<hipEvent_t> events;       // one per (stage, use) of the current call, created on first need, reused by later calls
    size_t events_used = 0;
    int take_event(hipEvent_t *out)
    {
        if (events_used == events.size()) {
            hipEvent_t e = nullptr;
            TTSAMD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            events.push_back(e);
        }
        *out = events[events_used++];
        return TTSAMD_OK;
    }
};

int add_conv(Model &m, const std::string &name, int c_out, int c_in, int kernel, int dilation)
{
    auto pc = std::make_unique<PackedConv>();
    RC(pack_named_conv(m.tensors, "hifigan", name, *pc, c_out, c_in, kernel, dilation));
    m.convs[name] = std::move(pc);
    return TTSAMD_OK;
}

// ConvTranspose1d weight [c_in, c_out, k] (stride u) -> the J-tap Conv1d [c_out * u, c_in, J], J = ceil(k / u), packed row
// co * u + r = phase r of output channel co: W'[m, ci, j'] = w[ci, co, r + (J - 1 - j') u] (taps >= k are zero); pad_left = J - 1
// (tts_amd/ops.py: convt_polyphase_weight)
int add_convt(Model &m, const std::string &name, int c_in, int c_out, int k, int u)
{
    HostTensor wt;
    int rc = fold_weight_norm(m.tensors, "hifigan", name, wt);
    if (rc) return rc;
    if (wt.shape.size() != 3 || wt.shape[0] != c_in || wt.shape[1] != c_out || wt.shape[2] != k) {
        set_error("hifigan: '%s' does not have the ConvTranspose1d shape [%d, %d, %d]", name.c_str(), c_in, c_out, k);
        return TTSAMD_ERR_INVALID;
    }
    const int J = (k + u - 1) / u;
    std::vector<float> w((size_t)c_out * u * c_in * J, 0.f);
    for (int co = 0; co < c_out; ++co)
        for (int r = 0; r < u; ++r)
            for (int ci = 0; ci < c_in; ++ci)
                for (int jp = 0; jp < J; ++jp) {
                    const int tap = r + (J - 1 - jp) * u;
                    if (tap < k) w[(((size_t)co * u + r) * c_in + ci) * J + jp] = wt.data[((size_t)ci * c_out + co) * k + tap];
                }
    const float *b = opt_bias(m.tensors, "hifigan", name, c_out, &rc);
    if (rc) return rc;
    std::vector<float> br;
    if (b) {
        br.resize((size_t)c_out * u);
        for (int co = 0; co < c_out; ++co)
            for (int r = 0; r < u; ++r) br[(size_t)co * u + r] = b[co];
    }
    auto pc = std::make_unique<PackedConv>();
    if ((rc = pack_conv(*pc, "hifigan", w.data(), b ? br.data() : nullptr, c_out * u, c_in, J, 1, J - 1))) return rc;
    m.convs[name] = std::move(pc);
    return TTSAMD_OK;
}

void fill_conv_args(const Model &m, ttsamd_conv1d_args &a, const PackedConv &pc, const float *x, int c_x, int t_in, float *y, int c_y, int t_y, int batch)
{
    model::fill_conv_args(m.precision, a, pc, x, c_x, t_in, y, c_y, t_y, batch);
}

// Activation workspace: a head (masks, padded input, conv_pre's output) and TWO stage arenas used alternately — after an upsample
// stage only its output is live, so stage i + 1 reuses the arena of stage i - 1.  The size pass (dry) records the peak of each arena;
// the real pass places arena 1 behind arena 0's peak.  (Before round 6 every stage took fresh memory: about 3x the live set.)
struct Workspace {
    unsigned char *base;
    size_t used = 0, cap;
    bool dry;              // size pass: nothing is launched
    size_t head = 0, arena_peak[2] = {0, 0}, arena_used[2] = {0, 0};
    int arena = -1;        // -1: the head
    void begin_stage(int i)
    {
        if (arena < 0) head = used;
        arena = i & 1;
        arena_used[arena] = 0;
    }
    float *take(size_t floats)
    {
        const size_t bytes = (floats * 4 + 255) & ~size_t(255);
        if (arena < 0) {
            float *p = dry ? nullptr : reinterpret_cast<float *>(base + used);
            used += bytes;
            return p;
        }
        const size_t off = head + (arena ? arena_peak[0] : 0) + arena_used[arena];
        arena_used[arena] += bytes;
        if (dry && arena_used[arena] > arena_peak[arena]) arena_peak[arena] = arena_used[arena];
        used = head + arena_peak[0] + arena_peak[1];
        return dry ? nullptr : reinterpret_cast<float *>(base + off);
    }
};

struct FuseLimits { int k128 = -1, k256 = -1; };
static FuseLimits fuse_limits_from_env()
{
    FuseLimits l;
    const char *e = getenv("TTSAMD_FUSE_LIMITS");
    for (; e && *e; ) {
        int c = 0, k = 0, n = 0;
        if (sscanf(e, "%d:%d%n", &c, &k, &n) != 2) break;
        if (c == 128) l.k128 = k;
        if (c == 256) l.k256 = k;
        e += n;
        if (*e == ',') ++e;
    }
    return l;
}

// which (channels, kernel) pairs the generator fuses (tts_amd/hifigan.py: fuse_channels, fuse_max_kernel = {128: 7 | 3, 256: 3 | 0})
bool fuse_pair(const Model &m, const PackedConv &c1, const PackedConv &c2, long cols)
{
    if (m.precision == 2) return false;
    const int ch = c1.c_out;
    if (!(c1.c_in == ch && c2.c_in == ch && c2.c_out == ch && c1.kernel == c2.kernel && c2.dilation == 1)) return false;
    // largest fused kernel size at 128 / 256 channels (tts_amd/hifigan.py: _fuse_limit, same table; TTSAMD_FUSE_LIMITS="128:7,256:7"
    // overrides it in both hosts for A/B runs).  256 channels: three products only, and not on small grids (cols = T x batch: the
    // 8-wave block is a large-grid tile)
    static const FuseLimits lim = fuse_limits_from_env();
    if (ch == 128 && c1.kernel > (lim.k128 >= 0 ? lim.k128 : (m.precision == 0 ? 7 : 3))) return false;
    if (ch == 256 && (m.precision != 0 || c1.kernel > (lim.k256 >= 0 ? lim.k256 : 7) || cols < 4096)) return false;
    return (m.precision == 0 ? ttsamd_resblock_pair_h2_supported(ch, c1.kernel, c1.dilation)
                             : ttsamd_resblock_pair_supported(ch, c1.kernel, c1.dilation)) != 0;
}

void fill_pair_args(const Model &m, ttsamd_resblock_args &r, const PackedConv &c1, const PackedConv &c2, const float *x, float *y, const float *accum,
                    const float *mask, int ch, int t, int batch, float div)
{
    memset(&r, 0, sizeof(r));
    const bool pad = ch < 32;
    r.x = x;
    r.y = y;
    r.accum = accum;
    r.mask = mask;
    r.w1_split = pad ? c1.w_split_pad32.p : c1.w_split.p;
    r.w2_split = pad ? c2.w_split_pad32.p : c2.w_split.p;
    r.w1_bytes = (int64_t)(pad ? c1.w_split_pad32.bytes : c1.w_split.bytes);
    r.w2_bytes = (int64_t)(pad ? c2.w_split_pad32.bytes : c2.w_split.bytes);
    if (m.precision == 0) {
        r.w1_h2 = pad ? c1.w_h2_pad32.p : c1.w_h2.p;
        r.w2_h2 = pad ? c2.w_h2_pad32.p : c2.w_h2.p;
        r.w1_h2_bytes = (int64_t)(pad ? c1.w_h2_pad32.bytes : c1.w_h2.bytes);
        r.w2_h2_bytes = (int64_t)(pad ? c2.w_h2_pad32.bytes : c2.w_h2.bytes);
    }
    r.bias1 = c1.has_bias ? static_cast<const float *>(c1.bias.p) : nullptr;
    r.bias2 = c2.has_bias ? static_cast<const float *>(c2.bias.p) : nullptr;
    r.c = ch;
    r.t = t;
    r.batch = batch;
    r.kernel = c1.kernel;
    r.dilation = c1.dilation;
    r.slope = kLreluSlope;
    r.out_div = div;
}

// a packed layer by name; a missing one is an error code, not a null dereference
const PackedConv *find_conv(const Model &m, const std::string &name)
{
    auto it = m.convs.find(name);
    if (it == m.convs.end() || !it->second) {
        set_error("hifigan: layer '%s' is not packed (ttsamd_hifigan_finalize did not complete)", name.c_str());
        return nullptr;
    }
    return it->second.get();
}
#define CONV(var, name)                          \
    const PackedConv *var##_p = find_conv(m, name); \
    if (!var##_p) return TTSAMD_ERR_INVALID;     \
    const PackedConv &var = *var##_p

// The launch sequence of HifiganGenerator.inference (hifigan_generator.py:267-282 -> forward, :236-265) on `st`; with ws.dry only the
// workspace size is computed.
int run(Model &m, Workspace &ws, const float *mel, int B, int T0, const int64_t *lengths, const float *in_mask, float *wav, hipStream_t st)
{
    const ttsamd_hifigan_config &c = m.cfg;
    const int p = c.inference_padding, nk = c.num_kernels, nu = c.num_upsamples;
    void *s = reinterpret_cast<void *>(st);
    m.events_used = 0;
    int T = T0 + 2 * p;
    // per-stage length masks of a ragged batch (one launch), then the replicate padding of every item's own frames
    std::vector<const float *> sm(nu + 1, nullptr);
    int64_t *len_eff = nullptr;
    if (lengths) {
        std::vector<int32_t> scales(nu + 1), ts(nu + 1);
        int sc = 1;
        size_t total = 0;
        for (int i = 0; i <= nu; ++i) {
            scales[i] = sc;
            ts[i] = T * sc;
            total += (size_t)B * ts[i];
            if (i < nu) sc *= c.upsample_factors[i];
        }
        float *masks = ws.take(total);
        len_eff = reinterpret_cast<int64_t *>(ws.take((size_t)B * 2));
        if (!ws.dry) {
            for (int s0 = 0; s0 <= nu; s0 += TTSAMD_MASK_MAX_STAGES) {
                const int n = std::min(TTSAMD_MASK_MAX_STAGES, nu + 1 - s0);
                size_t off = 0;
                for (int i = 0; i < s0; ++i) off += (size_t)B * ts[i];
                RC(ttsamd_stage_masks(masks + off, len_eff, lengths, B, 1, 2 * p, scales.data() + s0, ts.data() + s0, n, s));
            }
            size_t off = 0;
            for (int i = 0; i <= nu; ++i) {
                sm[i] = masks + off;
                off += (size_t)B * ts[i];
            }
        }
    }
    const float *x = mel;
    if (p > 0) {
        float *xp = ws.take((size_t)B * c.in_channels * T);
        if (!ws.dry) {
            if (lengths) RC(ttsamd_replicate_pad_ragged_ex(xp, mel, len_eff, -2 * p, B, c.in_channels, T0, p, s));
            else RC(ttsamd_replicate_pad(xp, mel, (int64_t)B * c.in_channels, T0, p, s));
        }
        x = xp;
    }
    int ch = c.upsample_initial_channel;
    float *o = ws.take((size_t)B * ch * T);
    ttsamd_conv1d_args a;
    CONV(cpre, "conv_pre");
    if (!ws.dry) {
        fill_conv_args(m, a, cpre, x, c.in_channels, T, o, ch, T, B);
        a.in_mask = lengths ? sm[0] : in_mask;        // in_mask: VITS feeds `z * y_mask` (vits.py:1161), multiplied in inside conv_pre's load
        RC(ttsamd_conv1d(&a, s));
    }
    for (int i = 0; i < nu; ++i) {
        const int u = c.upsample_factors[i], k_up = c.upsample_kernel_sizes[i];
        ch /= 2;
        const int pad_up = (k_up - u) / 2;
        const int T_up = (T - 1) * u - 2 * pad_up + k_up;
        ws.begin_stage(i);
        float *up = ws.take((size_t)B * ch * T_up);
        CONV(pu, "ups." + std::to_string(i));
        if (!ws.dry) {
            fill_conv_args(m, a, pu, o, ch * 2, T, up, ch, T_up, B);
            a.in_act = TTSAMD_ACT_LRELU;
            a.in_slope = kLreluSlope;
            a.mode = TTSAMD_CONV_SHUFFLE;
            a.shuffle_u = u;
            a.shuffle_pad = pad_up;
            a.in_mask = sm[i];
            a.t_out = T + pu.kernel - 1;
            if (pu.kernel != 2) a.w_h2 = nullptr, a.w_split = pu.w_split.p;     // polyphase forms other than two taps: the generic kernel
            RC(ttsamd_conv1d(&a, s));
        }
        const float *msk = sm[i + 1];
        T = T_up;
        float *o_next = ws.take((size_t)B * ch * T);
        float *zsum = nk > 1 ? ws.take((size_t)B * ch * T) : nullptr;
        // grouped stage (a single sentence): the branches of one ResBlock iteration as ONE launch, their outputs averaged by one more
        bool grouped = false;
        if (c.resblock_type == 1 && nk >= 2 && nk <= 3 && T % 4 == 0 && ttsamd_resblock_group_supported(ch, T, B)) {
            grouped = true;
            for (int j = 0; j < nk && grouped; ++j) {
                for (int d = 0; d < c.num_dilations[j]; ++d) {
                    if (c.num_dilations[j] != c.num_dilations[0] || c.resblock_dilation_sizes[j][d] != c.resblock_dilation_sizes[0][d]) grouped = false;
                }
                const int k = c.resblock_kernel_sizes[j];
                if (k != 3 && k != 7 && k != 11) grouped = false;
                for (int j2 = 0; j2 < j; ++j2)
                    if (c.resblock_kernel_sizes[j2] == k) grouped = false;
            }
            if (grouped) {
                for (int j = 0; j < nk && grouped; ++j)
                    for (int d = 0; d < c.num_dilations[0]; ++d) {
                        const std::string rp = "resblocks." + std::to_string(i * nk + j) + ".";
                        CONV(g1, rp + "convs1." + std::to_string(d));
                        CONV(g2, rp + "convs2." + std::to_string(d));
                        if (!fuse_pair(m, g1, g2, (long)T * B)) grouped = false;
                    }
            }
        }
        if (grouped) {
            std::vector<float *> buf(2 * nk);
            for (auto &b : buf) b = ws.take((size_t)B * ch * T);
            if (!ws.dry) {
                std::vector<const float *> cur(nk, up);
                for (int d = 0; d < c.num_dilations[0]; ++d) {
                    ttsamd_resblock_args arr[3];
                    memset(arr, 0, sizeof(arr));
                    for (int j = 0; j < nk; ++j) {
                        const std::string rp = "resblocks." + std::to_string(i * nk + j) + ".";
                        CONV(c1, rp + "convs1." + std::to_string(d));
                        CONV(c2, rp + "convs2." + std::to_string(d));
                        const int slot = c1.kernel == 3 ? 0 : (c1.kernel == 7 ? 1 : 2);
                        fill_pair_args(m, arr[slot], c1, c2, cur[j], buf[2 * j + (d & 1)], nullptr, msk, ch, T, B, 0.f);
                        arr[slot].w1_h2 = arr[slot].w2_h2 = nullptr;      // grouped launches run the six-product kernels
                        arr[slot].w1_h2_bytes = arr[slot].w2_h2_bytes = 0;
                    }
                    RC(ttsamd_resblock_group(arr, s));
                    for (int j = 0; j < nk; ++j) cur[j] = buf[2 * j + (d & 1)];
                }
                RC(ttsamd_sum_div(o_next, cur[0], cur[1], nk > 2 ? cur[2] : nullptr, (float)nk, (int64_t)B * ch * T, s));
            }
        } else {
            const bool side = m.concurrent && nk > 1;
            // (a branch on its own stream needs its own ping-pong buffers)
            std::vector<float *> xa(nk), xb(nk), tmp(nk);
            for (int j = 0; j < nk; ++j) {
                if (j == 0 || side) xa[j] = ws.take((size_t)B * ch * T), xb[j] = ws.take((size_t)B * ch * T), tmp[j] = ws.take((size_t)B * ch * T);
                else xa[j] = xa[0], xb[j] = xb[0], tmp[j] = tmp[0];
            }
            hipEvent_t ev_up = nullptr, prev_done = nullptr;
            if (side && !ws.dry) {
                for (int j = 0; j < nk; ++j)
                    if (!m.side[j]) TTSAMD_HIP(hipStreamCreateWithFlags(&m.side[j], hipStreamNonBlocking));
                RC(m.take_event(&ev_up));
                TTSAMD_HIP(hipEventRecord(ev_up, st));
            }
            for (int j = 0; j < nk && !ws.dry; ++j) {
                const std::string rp = "resblocks." + std::to_string(i * nk + j) + ".";
                const int nd = c.num_dilations[j];
                const float *cur = up;
                hipStream_t sj = side ? m.side[j] : st;
                void *s = reinterpret_cast<void *>(sj);
                if (side) TTSAMD_HIP(hipStreamWaitEvent(sj, ev_up, 0));
                for (int d = 0; d < nd; ++d) {
                    const bool last = d == nd - 1;
                    float *dst = last ? (j == nk - 1 ? o_next : zsum) : (cur == xa[j] ? xb[j] : xa[j]);
                    const float *accum = (last && j > 0) ? zsum : nullptr;
                    const float div = (last && j == nk - 1) ? (float)nk : 0.f;
                    const bool join = last && side && prev_done;      // zsum holds the previous branches' sum
                    if (c.resblock_type == 1) {
                        CONV(c1, rp + "convs1." + std::to_string(d));
                        CONV(c2, rp + "convs2." + std::to_string(d));
                        if (fuse_pair(m, c1, c2, (long)T * B)) {
                            if (join) TTSAMD_HIP(hipStreamWaitEvent(sj, prev_done, 0));
                            ttsamd_resblock_args r;
                            fill_pair_args(m, r, c1, c2, cur, dst, accum, msk, ch, T, B, div);
                            RC(ttsamd_resblock_pair(&r, s));
                        } else {
                            fill_conv_args(m, a, c1, cur, ch, T, tmp[j], ch, T, B);
                            a.in_act = TTSAMD_ACT_LRELU;
                            a.in_slope = kLreluSlope;
                            a.in_mask = msk;
                            RC(ttsamd_conv1d(&a, s));
                            if (join) TTSAMD_HIP(hipStreamWaitEvent(sj, prev_done, 0));
                            fill_conv_args(m, a, c2, tmp[j], ch, T, dst, ch, T, B);
                            a.in_act = TTSAMD_ACT_LRELU;
                            a.in_slope = kLreluSlope;
                            a.in_mask = msk;
                            a.res = cur;
                            a.res_bstride = (int64_t)ch * T;
                            a.res_rstride = T;
                            a.accum = accum;
                            a.accum_bstride = (int64_t)ch * T;
                            a.accum_rstride = T;
                            a.out_div = div;
                            RC(ttsamd_conv1d(&a, s));
                        }
                    } else {
                        CONV(cv, rp + "convs." + std::to_string(d));
                        if (join) TTSAMD_HIP(hipStreamWaitEvent(sj, prev_done, 0));
                        fill_conv_args(m, a, cv, cur, ch, T, dst, ch, T, B);
                        a.in_act = TTSAMD_ACT_LRELU;
                        a.in_slope = kLreluSlope;
                        a.in_mask = msk;
                        a.res = cur;
                        a.res_bstride = (int64_t)ch * T;
                        a.res_rstride = T;
                        a.accum = accum;
                        a.accum_bstride = (int64_t)ch * T;
                        a.accum_rstride = T;
                        a.out_div = div;
                        RC(ttsamd_conv1d(&a, s));
                    }
                    cur = dst;
                }
                if (side) {
                    RC(m.take_event(&prev_done));
                    TTSAMD_HIP(hipEventRecord(prev_done, sj));
                }
            }
            if (side && !ws.dry) TTSAMD_HIP(hipStreamWaitEvent(st, prev_done, 0));      // the last branch's final conv completes the chain
        }
        o = o_next;
    }
    CONV(cpost, "conv_post");
    if (!ws.dry) {
        // the final F.leaky_relu(o) uses the DEFAULT slope 0.01 (hifigan_generator.py:262), then conv_post and tanh
        fill_conv_args(m, a, cpost, o, ch, T, wav, c.out_channels, T, B);
        a.in_act = TTSAMD_ACT_LRELU;
        a.in_slope = 0.01f;
        a.out_act = TTSAMD_ACT_TANH;
        a.in_mask = sm[nu];
        RC(ttsamd_conv1d(&a, s));
    }
    return TTSAMD_OK;
}

void drop_graphs(Model &m)
{
    for (auto &g : m.graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    m.graphs.clear();
}

Model *as_model(void *h) { return static_cast<Model *>(h); }

// the launching pass over the handle's workspace, laid out as the size pass `dry` found it
Workspace real_ws(Model &m, const Workspace &dry)
{
    Workspace w{static_cast<unsigned char *>(m.work.p), 0, m.work.bytes, false};
    w.arena_peak[0] = dry.arena_peak[0];
    w.arena_peak[1] = dry.arena_peak[1];
    return w;
}

}  // namespace

extern "C" int ttsamd_hifigan_create(const ttsamd_hifigan_config *cfg, void **handle_out)
{
    return abi_guard("hifigan_create", [&]() -> int {
        TTSAMD_CHECK_ARG(cfg && handle_out, "hifigan_create: NULL argument");
        const ttsamd_hifigan_config &c = *cfg;
        TTSAMD_CHECK_ARG(c.in_channels > 0 && c.out_channels > 0 && c.upsample_initial_channel > 0, "hifigan_create: bad channel counts");
        TTSAMD_CHECK_ARG(c.resblock_type == 1 || c.resblock_type == 2, "hifigan_create: resblock_type must be 1 or 2");
        TTSAMD_CHECK_ARG(c.num_kernels >= 1 && c.num_kernels <= TTSAMD_HIFIGAN_MAX_KERNELS, "hifigan_create: 1..%d resblock kernels", TTSAMD_HIFIGAN_MAX_KERNELS);
        TTSAMD_CHECK_ARG(c.num_upsamples >= 1 && c.num_upsamples <= TTSAMD_HIFIGAN_MAX_UPSAMPLES, "hifigan_create: 1..%d upsample layers", TTSAMD_HIFIGAN_MAX_UPSAMPLES);
        TTSAMD_CHECK_ARG(c.inference_padding >= 0 && c.precision >= 0 && c.precision <= 2, "hifigan_create: bad padding / precision");
        int ch = c.upsample_initial_channel, hop = 1;
        for (int i = 0; i < c.num_upsamples; ++i) {
            TTSAMD_CHECK_ARG(c.upsample_factors[i] >= 1 && c.upsample_kernel_sizes[i] >= c.upsample_factors[i],
                             "hifigan_create: upsample layer %d: kernel %d < stride %d leaves samples without a tap", i, c.upsample_kernel_sizes[i], c.upsample_factors[i]);
            TTSAMD_CHECK_ARG(ch % 2 == 0, "hifigan_create: channel count %d cannot be halved at upsample layer %d", ch, i);
            ch /= 2;
            hop *= c.upsample_factors[i];
        }
        for (int j = 0; j < c.num_kernels; ++j) {
            TTSAMD_CHECK_ARG(c.num_dilations[j] >= 1 && c.num_dilations[j] <= TTSAMD_HIFIGAN_MAX_DILATIONS, "hifigan_create: resblock %d: 1..%d dilations", j, TTSAMD_HIFIGAN_MAX_DILATIONS);
            TTSAMD_CHECK_ARG(c.resblock_kernel_sizes[j] % 2 == 1, "hifigan_create: resblock kernel sizes are odd (get_padding, hifigan_generator.py:14-15)");
        }
        Model *m = new (std::nothrow) Model();
        TTSAMD_CHECK_ARG(m, "hifigan_create: out of memory");
        m->cfg = c;
        m->hop = hop;
        m->precision = c.precision;
        *handle_out = m;
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_hifigan_load(void *handle, const char *name, const float *data, const int64_t *shape, int ndim)
{
    return abi_guard("hifigan_load", [&]() -> int {
        TTSAMD_CHECK_ARG(handle, "hifigan_load: NULL handle");
        Model &m = *as_model(handle);
        RC(load_tensor(m.tensors, "hifigan", name, data, shape, ndim));
        m.finalized = false;
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_hifigan_finalize(void *handle)
{
    return abi_guard("hifigan_finalize", [&]() -> int {
        TTSAMD_CHECK_ARG(handle, "hifigan_finalize: NULL handle");
        Model &m = *as_model(handle);
        const ttsamd_hifigan_config &c = m.cfg;
        // a second finalize with nothing new loaded keeps the packed model (the host copies were dropped by the first one)
        if (m.finalized && m.tensors.empty()) return TTSAMD_OK;
        m.finalized = false;         // true again only when every layer has been packed
        // graphs and packed images of a previous weight set go first (a graph holds raw pointers to them)
        TTSAMD_HIP(hipDeviceSynchronize());
        drop_graphs(m);
        m.convs.clear();
        RC(add_conv(m, "conv_pre", c.upsample_initial_channel, c.in_channels, 7, 1));
        int ch = c.upsample_initial_channel;
        for (int i = 0; i < c.num_upsamples; ++i) {
            RC(add_convt(m, "ups." + std::to_string(i), ch, ch / 2, c.upsample_kernel_sizes[i], c.upsample_factors[i]));
            ch /= 2;
            for (int j = 0; j < c.num_kernels; ++j) {
                const std::string rp = "resblocks." + std::to_string(i * c.num_kernels + j) + ".";
                for (int d = 0; d < c.num_dilations[j]; ++d) {
                    if (c.resblock_type == 1) {
                        RC(add_conv(m, rp + "convs1." + std::to_string(d), ch, ch, c.resblock_kernel_sizes[j], c.resblock_dilation_sizes[j][d]));
                        RC(add_conv(m, rp + "convs2." + std::to_string(d), ch, ch, c.resblock_kernel_sizes[j], 1));
                    } else {
                        RC(add_conv(m, rp + "convs." + std::to_string(d), ch, ch, c.resblock_kernel_sizes[j], c.resblock_dilation_sizes[j][d]));
                    }
                }
            }
        }
        RC(add_conv(m, "conv_post", c.out_channels, ch, 7, 1));
        m.tensors.clear();           // the host copies are not needed any more
        m.finalized = true;
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_hifigan_set_option(void *handle, int option, int value)
{
    return abi_guard("hifigan_set_option", [&]() -> int {
        TTSAMD_CHECK_ARG(handle, "hifigan_set_option: NULL handle");
        Model &m = *as_model(handle);
        switch (option) {
            case TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES:
                if (m.concurrent != (value != 0)) {
                    TTSAMD_HIP(hipDeviceSynchronize());
                    drop_graphs(m);                 // a captured sequence has its branch topology baked in
                    m.concurrent = value != 0;
                }
                return TTSAMD_OK;
        }
        set_error("hifigan_set_option: unknown option %d", option);
        return TTSAMD_ERR_INVALID;
    });
}

extern "C" int64_t ttsamd_hifigan_output_samples(void *handle, int frames)
{
    if (!handle || frames < 0) return -1;
    Model &m = *as_model(handle);
    int64_t T = frames + 2 * m.cfg.inference_padding;
    for (int i = 0; i < m.cfg.num_upsamples; ++i)
        T = (T - 1) * m.cfg.upsample_factors[i] - 2 * ((m.cfg.upsample_kernel_sizes[i] - m.cfg.upsample_factors[i]) / 2) + m.cfg.upsample_kernel_sizes[i];
    return T;
}

extern "C" int ttsamd_hifigan_forward(void *handle, const float *mel, int batch, int frames, const int64_t *lengths, float *wav, int use_graph,
                                      void *stream)
{
    return ttsamd_hifigan_forward_ex(handle, mel, batch, frames, lengths, nullptr, wav, use_graph, stream);
}

extern "C" int ttsamd_hifigan_forward_ex(void *handle, const float *mel, int batch, int frames, const int64_t *lengths, const float *in_mask,
                                         float *wav, int use_graph, void *stream)
{
    return abi_guard("hifigan_forward", [&]() -> int {
        TTSAMD_CHECK_ARG(handle && mel && wav, "hifigan_forward: NULL argument");
        TTSAMD_CHECK_ARG(!(lengths && in_mask), "hifigan_forward: lengths (ragged-exact batching) and in_mask are alternatives");
        TTSAMD_CHECK_ARG(!in_mask || as_model(handle)->cfg.inference_padding == 0, "hifigan_forward: in_mask is the decoder-inside-VITS form (inference_padding 0)");
        Model &m = *as_model(handle);
        TTSAMD_CHECK_ARG(m.finalized, "hifigan_forward: weights not loaded (ttsamd_hifigan_load ... ttsamd_hifigan_finalize)");
        TTSAMD_CHECK_ARG(batch >= 0 && frames >= 1 && batch <= 65535, "hifigan_forward: bad shape");
        if (batch == 0) return TTSAMD_OK;
        if (lengths) {
            for (int i = 0; i < m.cfg.num_upsamples; ++i)
                TTSAMD_CHECK_ARG((m.cfg.upsample_kernel_sizes[i] - m.cfg.upsample_factors[i]) % 2 == 0,
                                 "hifigan_forward: ragged batching needs upsample kernels with k - stride even (output = frames * hop exactly)");
        }
        hipStream_t st = as_stream(stream);
        Workspace dry{nullptr, 0, 0, true};
        RC(run(m, dry, mel, batch, frames, lengths, in_mask, wav, st));
        if (dry.used > m.work.bytes) {
            // growing the workspace invalidates every captured graph (they hold pointers into it)
            TTSAMD_HIP(hipDeviceSynchronize());
            drop_graphs(m);
            if (m.work.p) TTSAMD_HIP(hipFree(m.work.p));
            m.work.p = nullptr;
            m.work.bytes = 0;
            TTSAMD_HIP(hipMalloc(&m.work.p, dry.used));
            m.work.bytes = dry.used;
        }
        if (use_graph) {
            for (auto &g : m.graphs)
                if (g.mel == mel && g.lengths == lengths && g.in_mask == in_mask && g.wav == wav && g.batch == batch && g.frames == frames && g.stream == st) {
                    TTSAMD_HIP(hipGraphLaunch(g.exec, st));
                    return TTSAMD_OK;
                }
            // first sighting of this (buffers, shape, stream): run it once eagerly (one-time function attributes are set outside any
            // capture), then capture the same sequence and replay from the next call on
            Workspace w0 = real_ws(m, dry);
            RC(run(m, w0, mel, batch, frames, lengths, in_mask, wav, st));
            VocGraph e{mel, lengths, in_mask, wav, batch, frames, st};
            if (!m.cap_stream) TTSAMD_HIP(hipStreamCreateWithFlags(&m.cap_stream, hipStreamNonBlocking));
            TTSAMD_HIP(hipStreamBeginCapture(m.cap_stream, hipStreamCaptureModeThreadLocal));
            Workspace w1 = real_ws(m, dry);
            const int rc = run(m, w1, mel, batch, frames, lengths, in_mask, wav, m.cap_stream);
            const hipError_t he = hipStreamEndCapture(m.cap_stream, &e.graph);
            if (rc) {
                if (e.graph) (void)hipGraphDestroy(e.graph);
                return rc;
            }
            if (he != hipSuccess) {
                if (e.graph) (void)hipGraphDestroy(e.graph);
                TTSAMD_HIP(he);
            }
            const hipError_t hi = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
            if (hi != hipSuccess) {
                (void)hipGraphDestroy(e.graph);
                TTSAMD_HIP(hi);
            }
            if (m.graphs.size() >= 16) {
                VocGraph &old = m.graphs.front();
                // the oldest entry's stream may have been destroyed by its owner since: then wait for the whole device instead
                if (hipStreamSynchronize(old.stream) != hipSuccess) {
                    (void)hipGetLastError();
                    (void)hipDeviceSynchronize();
                }
                (void)hipGraphExecDestroy(old.exec);
                (void)hipGraphDestroy(old.graph);
                m.graphs.erase(m.graphs.begin());
            }
            m.graphs.push_back(e);
            return TTSAMD_OK;       // (the eager run above produced this call's result)
        }
        Workspace w = real_ws(m, dry);
        return run(m, w, mel, batch, frames, lengths, in_mask, wav, st);
    });
}

extern "C" int ttsamd_hifigan_destroy(void *handle)
{
    return abi_guard("hifigan_destroy", [&]() -> int {
        if (!handle) return TTSAMD_OK;
        Model *m = as_model(handle);
        (void)hipDeviceSynchronize();
        drop_graphs(*m);
        if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
        for (auto &sd : m->side)
            if (sd) (void)hipStreamDestroy(sd);
        for (auto e : m->events) (void)hipEventDestroy(e);
        delete m;
        return TTSAMD_OK;
    });
}
