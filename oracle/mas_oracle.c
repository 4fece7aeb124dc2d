/* TEST INFRASTRUCTURE ONLY — CPU oracle for monotonic alignment search (MAS).
 *
 * Plain-C restatement of the reference's only native component:
 *   TTS/tts/utils/monotonic_align/core.pyx:11-37  maximum_path_each
 *   TTS/tts/utils/monotonic_align/core.pyx:42-47  maximum_path_c
 * Pinned (tests/test_mas_oracle.py) bit-for-bit against the reference's own
 * Cython module compiled from /root/reference into oracle/_ref/ and against
 * helpers.maximum_path_numpy (TTS/tts/utils/helpers.py:197-236).
 *
 * Never linked into libtts_amd.so; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call it.
 */
#include <stdint.h>

static inline float fmax_(float a, float b) { return a > b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }
static inline int imin_(int a, int b) { return a < b ? a : b; }

/* core.pyx:11-37. `path` and `value` are row-major [t_x_stride rows][row_stride]. */
void mas_oracle_each(int32_t *path, float *value, int t_x, int t_y, int row_stride, float max_neg_val)
{
    int index = t_x - 1;
    for (int y = 0; y < t_y; ++y) {                                         /* core.pyx:19 */
        for (int x = imax_(0, t_x + y - t_y); x < imin_(t_x, y + 1); ++x) { /* core.pyx:20 */
            float v_cur, v_prev;
            if (x == y) v_cur = max_neg_val;                                /* core.pyx:21-24 */
            else        v_cur = value[(long)x * row_stride + (y - 1)];
            if (x == 0) v_prev = (y == 0) ? 0.f : max_neg_val;              /* core.pyx:25-31 */
            else        v_prev = value[(long)(x - 1) * row_stride + (y - 1)];
            value[(long)x * row_stride + y] = fmax_(v_cur, v_prev) + value[(long)x * row_stride + y]; /* :32 */
        }
    }
    for (int y = t_y - 1; y >= 0; --y) {                                    /* core.pyx:34-37 */
        path[(long)index * row_stride + y] = 1;
        if (index != 0 && (index == y ||
                           value[(long)index * row_stride + (y - 1)] < value[(long)(index - 1) * row_stride + (y - 1)]))
            index = index - 1;
    }
}

/* core.pyx:42-47: batch driver (the reference's prange is serial: no OpenMP flags in setup.py:74-79). */
void mas_oracle_c(int32_t *paths, float *values, const int32_t *t_xs, const int32_t *t_ys,
                  int b, int t_x_max, int t_y_max, float max_neg_val)
{
    for (int i = 0; i < b; ++i)
        mas_oracle_each(paths + (long)i * t_x_max * t_y_max, values + (long)i * t_x_max * t_y_max,
                        t_xs[i], t_ys[i], t_y_max, max_neg_val);
}
