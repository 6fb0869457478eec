/* oracle/frontend_oracle.h -- TEST INFRASTRUCTURE ONLY (see frontend_oracle.c). */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_PIX_RGB24 = 0, ORC_PIX_YUV420P = 1 };

/* one separable filter bank: dst positions x taps; left[i] = first source index of position i (may be < 0 / run past
 * the end: source indices are clamped), coef[i * taps + k] 14-bit fixed point, every row sums to 16384 */
int orc_scaler_filter(int src, int dst, int *taps, int32_t *left, int16_t *coef, int cap);

/* src: one picture (RGB24: rows of 3 * src_w bytes; YUV420P: Y, U, V planes back to back); out: NV21, dst_w * dst_h * 3 / 2 */
int orc_scaler_convert(int src_format, int src_w, int src_h, int src_full_range, int dst_w, int dst_h,
                       const uint8_t *src, uint8_t *out);

#ifdef __cplusplus
}
#endif
