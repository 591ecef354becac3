/* TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Serial C restatement of the CPU kernel of the third-party library torch_scatter 2.0.x
 * (csrc/cpu/scatter_cpu.cpp + csrc/cpu/reducer.h), specialised to the only layout the ptgnn hot
 * path uses (abstractmessagepassing.py:44-50): src [E, D] row-major fp32, index int64 [E]
 * broadcast along dim 0, out [N, D].
 *
 *   reduce: 0 = sum, 1 = mean, 2 = max, 3 = min, 4 = mul (init 1, product in edge order, empty segments stay 1)
 *   out is initialised to 0 (sum/mean) or the dtype's lowest/highest (max/min); elements are
 *   folded in edge order; mean divides by max(count, 1); for max/min, segments that received no
 *   element are written as 0 and arg (nullable) = E for them, else the winning edge position.
 *
 * The library is not vendored under /root/reference; this file exists so that the torch-based
 * restatement (oracle/scatter_ref.py) is cross-checked by an independent implementation.
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>

int ptgnn_oracle_scatter_f32(const float *src, const int64_t *index, int64_t E, int64_t D,
                             int64_t N, int reduce, float *out, int64_t *arg) {
  if (reduce < 0 || reduce > 4) return -1;
  const float init = reduce == 2 ? -FLT_MAX : (reduce == 3 ? FLT_MAX : (reduce == 4 ? 1.0f : 0.0f));
  for (int64_t i = 0; i < N * D; ++i) out[i] = init;
  if (arg) for (int64_t i = 0; i < N * D; ++i) arg[i] = E;
  int64_t *count = NULL;
  if (reduce == 1) count = (int64_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int64_t));
  for (int64_t e = 0; e < E; ++e) {
    const int64_t v = index[e];
    if (v < 0 || v >= N) { free(count); return -2; }
    float *o = out + v * D;
    const float *s = src + e * D;
    if (reduce <= 1) {
      for (int64_t d = 0; d < D; ++d) o[d] += s[d];
      if (count) count[v]++;
    } else if (reduce == 4) {
      for (int64_t d = 0; d < D; ++d) o[d] *= s[d];
    } else if (reduce == 2) {
      for (int64_t d = 0; d < D; ++d)
        if (s[d] > o[d]) { o[d] = s[d]; if (arg) arg[v * D + d] = e; }
    } else {
      for (int64_t d = 0; d < D; ++d)
        if (s[d] < o[d]) { o[d] = s[d]; if (arg) arg[v * D + d] = e; }
    }
  }
  if (reduce == 1) {
    for (int64_t v = 0; v < N; ++v) {
      const float c = (float)(count[v] < 1 ? 1 : count[v]);
      for (int64_t d = 0; d < D; ++d) out[v * D + d] /= c;
    }
    free(count);
  } else if (reduce == 2 || reduce == 3) {
    /* torch_scatter: out.masked_fill_(arg_out == src.size(dim), 0) */
    char *touched = (char *)calloc((size_t)(N > 0 ? N : 1), 1);
    for (int64_t e = 0; e < E; ++e) touched[index[e]] = 1;
    for (int64_t v = 0; v < N; ++v)
      if (!touched[v]) for (int64_t d = 0; d < D; ++d) out[v * D + d] = 0.0f;
    free(touched);
  }
  return 0;
}
