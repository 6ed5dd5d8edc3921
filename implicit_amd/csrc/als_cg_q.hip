// K1q: dispatch of the CG half sweep for f = 64 / 128 over the row-length classes of the schedule (imp_csr::bin_start).
//
// Arithmetic contract: the oracle's CG (implicit/cpu/_als.pyx:152-248).  Kernels -- all on quarter-layout register tiles
// (als_qtile.h), the whole row resident for all 1 + cg_steps passes:
//   short rows (<= 32 nnz)   f = 128: one wavefront per row, 16 rows per workgroup in lock step, the gramian product on the bf16
//                            matrix cores (als_cg_qf.hip als_cg_qfgroup_kernel);  f = 64: independent wavefronts (team width 1)
//   mid rows (33 .. 512)     a team of 2 / 4 / 8 / 16 wavefronts per row, leader protocol (als_cg_qf.hip als_cg_qfteam_kernel);
//                            float16 storage: packed 64-entry tiles, half the wavefronts per row (als_cg_qh.hip)
// (The first generation of these kernels -- every wavefront of a team repeating the CG arithmetic, rounds 1-2 -- and the
// IMP_TEAM_FUSED / IMP_SHORT_TEAM1 / IMP_TILE64 switches that selected them were removed in round 6.)
#include <type_traits>

#include "als_qtile.h"
#include "common.h"

namespace imp {

// als_cg_qf.hip
template <typename T>
void launch_team_fused(const imp_csr *C, int f, int width, int first, int count, T *X, const T *Y, const float *A0, int cg_steps,
                       const char *name);
template <typename T>
void launch_group_fused(const imp_csr *C, int f, int first, int count, T *X, const T *Y, const float *A0, int cg_steps, const char *name);
// als_cg_qh.hip: 64-entry tiles of packed halves, half the wavefronts per row (float16 storage)
template <typename ST>
void launch_team_tile64(const imp_csr *C, int f, int width, int first, int count, ST *X, const ST *Y, const float *A0, int cg_steps,
                        const char *name);

template <int F, typename T> static void run_classes_q(const imp_csr *C, T *X, const T *Y, const float *A0, int cg_steps) {
  const int32_t *b = C->bin_start;  // classes: 1 (256,512]  2 (128,256]  3 (64,128]  4 (32,64]  5 (16,32]  6 (0,16]
  constexpr bool kHalf = std::is_same<T, __half>::value;
  // team width per class: a wavefront holds 32 entries (fp32 storage) or 64 (packed halves)
  auto team = [&](int width32, int lo, int hi, const char *name) {
    if constexpr (kHalf) launch_team_tile64<T>(C, F, width32 / 2, lo, hi - lo, X, Y, A0, cg_steps, name);
    else launch_team_fused<T>(C, F, width32, lo, hi - lo, X, Y, A0, cg_steps, name);
  };
  team(16, b[1], b[2], "als_cg_team16_rows");  // (names: the row class)
  team(8, b[2], b[3], "als_cg_team8_rows");
  team(4, b[3], b[4], "als_cg_team4_rows");
  team(2, b[4], b[5], "als_cg_team2_rows");
  // short rows.  f = 128: 16 rows per workgroup in lock step with the gramian product on the matrix cores (measured 1.16 ms per
  // configs[2] iteration against 1.38 ms for independent waves with the VALU product -- at f = 128 the product is 57 % of a short
  // row's arithmetic); f = 64: independent waves win (configs[1] shape: 0.84 against 0.94 ms), the product is a quarter of the size
  if constexpr (F == 64) launch_team_fused<T>(C, F, 1, b[5], b[7] - b[5], X, Y, A0, cg_steps, "als_cg_short_rows");
  else launch_group_fused<T>(C, F, b[5], b[7] - b[5], X, Y, A0, cg_steps, "als_cg_short_rows");
}

template <typename T> void least_squares_cg_q(const imp_csr *C, T *X, const T *Y, const float *A0, int f, int cg_steps) {
  if (f == 128) run_classes_q<128, T>(C, X, Y, A0, cg_steps);
  else if (f == 64) run_classes_q<64, T>(C, X, Y, A0, cg_steps);
  else throw std::invalid_argument("least_squares_cg_q: f must be 64 or 128");
}
template void least_squares_cg_q<float>(const imp_csr *, float *, const float *, const float *, int, int);
template void least_squares_cg_q<__half>(const imp_csr *, __half *, const __half *, const float *, int, int);

}  // namespace imp
