// fft_configs.hpp -- the supported transform lengths and their geometry.
// Length L, T threads per transform (T == 64: one wavefront per transform, no
// workgroup barriers inside a transform), forward radix list (the inverse runs
// it reversed); C = spectrum columns per workgroup in the column kernel and
// Q = row pairs per workgroup in the row kernels, for f32 / f64.
// LDS per workgroup = C * (L*9/8 + 1) * sizeof(complex).
#pragma once
#include "fft_core.hpp"

namespace rl {

template <int L>
struct CfgFor;

template <>
struct CfgFor<64> {   // unit-test size
    using Cfg = FftCfg<64, 8, 8, 8>;
    static constexpr int C32 = 8, C64 = 8, Q32 = 8, Q64 = 8;
};
template <>
struct CfgFor<192> {  // 128 + 53
    // 16 lanes per transform, four transforms per wave: every lane has a butterfly in the
    // radix-12 pass and 12 of 16 in the radix-16 pass, against 24 of 64 in the radix-8 passes of
    // (3,8,8) on a whole wave.  Measured at 128x128, f32: +25 % (one view), +8 % (four views, which
    // lose the fused multi-view column modes that exist for T == 64 only).  Row kernels still
    // exchange through LDS with wave-level ordering only (WaveLocal).
    using Cfg = FftCfg<192, 16, 12, 16>;
    // 16 columns (128-B row segments, 256 threads): +10 % over 32 at 128x128, 8 measures the same
    static constexpr int C32 = 16, C64 = 16, Q32 = 16, Q64 = 16;
};
template <>
struct CfgFor<256> {  // 160 + 53
    // wave-private; (16,16) on 16 lanes measures +14 % for one view but -9 % for four (no fused
    // multi-view column modes), so the whole-wave geometry stays
    using Cfg = FftCfg<256, 64, 4, 8, 8>;
    static constexpr int C32 = 8, C64 = 4, Q32 = 4, Q64 = 4;
};
template <>
struct CfgFor<576> {  // 512 + 53  (the BASELINE headline size)
    // wave-private; radix 9 first so that the row kernels' pointwise stage (inverse's
    // last = forward's first pass) holds elements lane + 64 r: fully coalesced rows
    using Cfg = FftCfg<576, 64, 9, 8, 8>;
#ifndef RL_576_C32
#define RL_576_C32 8      // columns (= waves) per column workgroup, f32 (round 4 A/B lever: 16 = whole 128-byte lines, two 16-wave workgroups per CU)
#endif
    static constexpr int C32 = RL_576_C32, C64 = 4, Q32 = 4, Q64 = 4;
};
template <>
struct CfgFor<1152> { // 1024 + 53
    using Cfg = FftCfg<1152, 144, 8, 9, 16>;
    // one row pair per workgroup: the row kernels of the workgroup-synchronous lengths have barriers
    // between passes, and many small workgroups overlap them (1024^2: 2525 -> 3108 frames/s over Q = 4)
    static constexpr int C32 = 4, C64 = 4, Q32 = 1, Q64 = 1;   // T*C <= 1024 threads
};
template <>
struct CfgFor<2304> { // 2048 + 53
    // (12,12,16) on 192 threads keeps every lane busy in two of three passes but measures the same
    // (473 vs 478 frames/s at 2048^2): the 4-column tiles (32-B row segments) bound the column pass
    using Cfg = FftCfg<2304, 256, 9, 16, 16>;   // row kernels; the column kernels use ColCfgFor<2304> below
    // one row pair per workgroup: 7 workgroups per CU overlap their phases (+9 % at 2048^2 over 2)
    // f64 column tiles: 2 / 3 columns 225 / 252 frames/s at 2048^2 (4 do not fit the LDS)
    static constexpr int C32 = 6, C64 = 3, Q32 = 1, Q64 = 1;
};

template <>
struct CfgFor<4608> { // 4096 + 53
    using Cfg = FftCfg<4608, 576, 8, 8, 8, 9>;     // row kernels; columns: ColCfgFor<4608> below
    // f64 column tiles: 1 / 2 columns 35.2 / 46.5 frames/s at 4096^2, K = 20
    static constexpr int C32 = 3, C64 = 2, Q32 = 1, Q64 = 1;
};

// One pad slot per 16 elements for L = 2304 (3): 4 columns then take 78 KB instead
// of 83 KB and two column workgroups fit a CU.
template <>
struct LdsPadShift<2304> {
    static constexpr int value = 3;
};

// LDS layout per length: the gathered exchange layout (fft_core.hpp, LdsGather) for the
// wave-private geometries
template <>
struct LdsGather<256> {
    static constexpr bool value = true;
};
template <>
struct LdsGather<576> {
    static constexpr bool value = true;
};
// L = 576: the first forward exchange (radix 9, 64 butterflies) is conflict free at stride 89 only -- 801 slots = 6.4 KB per f32
// transform, three 8-wave workgroups per CU, and every f32 RL kernel of this length needs <= 64 registers since round 4, i.e.
// could keep 8 waves per SIMD.  Capped at 608 slots (stride 66: one ds_read_b64 of the pass takes 4 LDS cycles instead of 2)
// a transform is 610 slots = 4.9 KB: four workgroups per CU.
#ifndef RL_LDS_CAP_576
#define RL_LDS_CAP_576 608
#endif
template <>
struct LdsMaxSlots<576> {
    static constexpr int value = RL_LDS_CAP_576;
};
    // measured on the row kernels alone: 2048^2 -2 ... -3 %, 4096^2 ROW_RATIO -16 %, ROW_UPDATE -4 %
// the long, workgroup-synchronous lengths (row kernels; f64 column kernels): 1
template <>
struct LdsGather<1152> {
    static constexpr bool value = true;
};
template <>
struct LdsGather<2304> {
    static constexpr bool value = true;
};
template <>
struct LdsGather<4608> {
    static constexpr bool value = true;
};

// Geometry of the COLUMN kernels where it differs from the row kernels' (same length, its own radix
// list, thread count and twiddle table).  A column workgroup holds C whole columns in LDS, so few
// threads per transform = more columns per workgroup = wider row segments of the tile (C * 8 bytes);
// the row kernels want the opposite, many threads per transform and the radix that touches the
// image rows first.  Measured per 512^2-equivalent frame:
//   2304: rows 0.98 / 1.24 us on (9,16,16) x 256 threads against 1.79 / 1.65 on (16,16,9) x 144;
//         columns 2.6 us on the former (4 columns, 32-B segments), 2.26 on the latter (6 columns);
//   4608: rows 1.2 / 1.7 us on (8,8,8,9) x 576 against 1.5 / 2.2 on (16,16,18) x 288;
//         columns 6.1 us on the former (1 column, 8-B segments), 3.8 on the latter (3 columns).
template <int L>
struct ColCfgFor {
    using type = typename CfgFor<L>::Cfg;
};
template <>
struct ColCfgFor<2304> {
    using type = FftCfg<2304, 144, 16, 16, 9>;
};
template <>
struct ColCfgFor<4608> {
    using type = FftCfg<4608, 288, 16, 16, 18>;
};

// Long column transforms done as M wave-private core transforms plus one outer radix-M step in registers
// (conv_kernels.hpp colconv_outer_body), f32 only.  2304 = 4 x 576: LDS holds one residue class of the
// tile (8 columns x 576 rows = the L = 576 kernel's 51 KB) instead of whole 2304-row columns (124 KB for
// 6 columns, one workgroup per CU).
template <int L>
struct OuterCol {
    static constexpr bool value = false;
    using Core = typename CfgFor<64>::Cfg;   // unused
    static constexpr int M = 2, C = 8, CW = 8, MIN_WAVES = 1;   // CW: columns per workgroup of the whole pass (C: of the split pass)
    static constexpr bool SPLIT = false;                 // the split pass (COL_SPLIT_*) is what multi-view f32 plans run
    static constexpr int PARK = 0;                       // waiting core results per lane the whole pass keeps in LDS (colconv_outer_body)
    static constexpr int TWLDS = 0;                      // the whole pass reads the core's (1) and the outer (2) twiddles from an LDS copy
    static constexpr int TWLDS_SPLIT = 0;                // the same for the halves of the split pass
    // float64 (round 4): the whole pass on the same body -- C64 columns (= waves) per workgroup, PARK64 waiting values per lane in
    // LDS, MIN_WAVES64 waves per SIMD for the register budget; no twiddle copies (the float64 core table is 36 KB), no split pass
    static constexpr bool value64 = false;
    static constexpr int C64 = 4, PARK64 = 0, MIN_WAVES64 = 2;
};
// Multi-view plans on these lengths run the SPLIT pass (conv_kernels.hpp COL_SPLIT_*).  History of the fused multi-view modes
// here: on colconv_outer_body two 4 x 10 register sets (256 VGPRs + 81 spilled dwords: round 2); on a two-waves-per-column body
// 128 VGPRs + 60-100 spilled dwords, 991 us against 730 us for V per-image launches; on a four-waves-per-column body with the
// ratio spectra in a 4 x 4 blocked layout no spill and 205-218 frames/s at 2048^2 x 4 views (V per-image launches: 190) -- the
// split pass gives 236-244.  All removed.
template <>
struct OuterCol<2304> {
    static constexpr bool value = true;
    using Core = typename CfgFor<576>::Cfg;
    // columns (= waves) per workgroup of the outer-decimation column kernel.  Measured at 2048^2,
    // 16-frame launches: 4 / 8 / 16 columns 565 / 611 / 599 us, one or two workgroups per CU
    // (4 2 / 4) 569 / 611 us: the kernel's time does not move with its geometry
    static constexpr int M = 4, C = 8, MIN_WAVES = 4;   // waves per SIMD the register budget is cut for
    // The whole pass (single-view plans) on 16-column tiles: whole 128-byte lines, ONE 16-wave workgroup per CU -- and then both
    // twiddle tables fit beside the transforms (102.6 + 15.8 + 13.5 + 24.6 KB of parking space).  Measured, 2048^2 point, one box:
    // 8 columns / core table 741 frames/s, 16 / core table 745, 16 / both tables 771.  The split pass stays on 8 columns
    // (2048^2 x 4 views 248 against 239-247, x 2 views 425 against 404-417).
    static constexpr int CW = 16;
    static constexpr bool SPLIT = true;    // 2048^2: 4 views 215 -> 236-244 frames/s, 2 views 407 -> 414
#ifndef RL_PARK_2304
#define RL_PARK_2304 3
#endif
    // (history, 8-column tiles at two workgroups per CU: 28.7 KB each beside the transforms -- 7 parked values and no copy 664-688
    // frames/s at 2048^2 point, the core's table + 3 parked values 766, the table alone 760)
    static constexpr int TWLDS = 2;
    static constexpr int TWLDS_SPLIT = 1;   // 2 x (51.3 + 15.8) KB; with the outer table 2 x 80.9 KB would not fit
    static constexpr int PARK = RL_PARK_2304;
    // float64: 4 x 10 complex doubles per lane wait for the radix-4 step -- 160 registers beside a core transform's ~110 -- so 10 of
    // them wait in LDS; 4 columns per workgroup (64-byte row segments, as the f32 kernel's 8), two 4-wave workgroups per CU at up to
    // 256 registers: 2 x (39 + 41) KB.  (rounds 1-3: the workgroup-synchronous (16,16,9) x 144 kernel on 3-column tiles)
#ifndef RL_PARK64_2304
#define RL_PARK64_2304 10
#endif
    static constexpr bool value64 = value;
    static constexpr int C64 = 4, PARK64 = RL_PARK64_2304, MIN_WAVES64 = 2;
};
// 1152 = 2 x 576 (round 3, for the split pass of multi-view plans; as a whole-pass kernel it measured 1.32 -> 1.14 us alone and
// no gain in the 1024^2 single-view loop in round 2)
template <>
struct OuterCol<1152> {
    static constexpr bool value = true;
    using Core = typename CfgFor<576>::Cfg;
    // (16: 1024^2 x 4 views 1084 -> 1030 frames/s, point 3900 -> 3820; 9 -- 128 tiles per pair spectrum,
    // a 4-pair launch exactly one round of workgroups -- 897 / 3560; 12: 972 / 3758: unaligned segments cost more)
    // (6 = three workgroups per CU at 80 registers + 48 bytes of scratch: 1024^2 x 4 views 1025 -> 890 frames/s)
    static constexpr int M = 2, C = 8, CW = 8, MIN_WAVES = 4;
    // measured at 1024^2 (frames/s; (8,9,16) x 144 workgroup-synchronous kernel / this body per image / its split pass):
    // 4 views 705 / 1015-1043 / 995-1027, 2 views - / 1900 / 1690-1740, 1 view (frame pairs) - / 3820-3850 / -;
    // with the twiddle copies in LDS: 4 views 1078 per image / 993 split, 2 views 1966 / 1650 (not used at this length)
    static constexpr bool SPLIT = false;
    static constexpr int PARK = 0;   // (2 x 10 values per lane: nothing spills)
    static constexpr int TWLDS = 2;   // 2 x (51.3 + 15.8 + 4.5) KB
    static constexpr int TWLDS_SPLIT = 2;
    static constexpr bool value64 = value;   // float64: 2 x 10 complex doubles per lane, nothing parked
    static constexpr int C64 = 4, PARK64 = 0, MIN_WAVES64 = 2;
};
// 4608 = 8 x 576 on the same body: 8 x 10 complex values wait in registers.  Measured (us per 512^2-equivalent frame,
// column kernel alone; whole 20-iteration loop): 3.37 -> 1.96, 4096^2 loop 23.6 -> 17.8 ms per 2 frames.
// (1152 = 2 x 576 was built too: 1.32 -> 1.14 us alone, no gain in the 1024^2 loop; the (8,9,16) x 144 kernel stays.)
template <>
struct OuterCol<4608> {
    static constexpr bool value = true;
    using Core = typename CfgFor<576>::Cfg;
    static constexpr int M = 8, C = 8, CW = 8, MIN_WAVES = 2;   // one 8-wave workgroup per CU, 256 registers per lane
    static constexpr bool SPLIT = true;    // 4096^2, 4 views: 33.5 -> 45 frames/s
#ifndef RL_PARK_4608
#define RL_PARK_4608 14
#endif
    static constexpr int TWLDS = 2;   // + 15.8 KB (core) + 31.5 KB (outer)
    static constexpr int TWLDS_SPLIT = 2;
    // float64 (end of round 4): 8 x 10 complex doubles per lane = 320 registers -- 28 of them wait in LDS, ONE 4-wave workgroup per CU
    // at ~415 registers (accumulation registers included), 4 x 9.7 KB of transforms + 112 KB of parking space; the workgroup-
    // synchronous (16,16,18) x 288 kernel it replaces: 66.9 -> 58.5 ms per 4 frames x 21 passes at 4096^2 (parked 20 / 24 / 28:
    // 65.96 / 65.60 / 64.88 ms before the tile I/O change, conv_kernels.hpp WIDE)
#ifndef RL_PARK64_4608
#define RL_PARK64_4608 28
#endif
    static constexpr bool value64 = value;
    static constexpr int C64 = 4, PARK64 = RL_PARK64_4608, MIN_WAVES64 = 1;
    static constexpr int PARK = RL_PARK_4608;   // one workgroup per CU: 57 KB of parking space beside 39 KB of transforms (round 4: 10 values left 40-52 bytes of scratch per lane at HEAD, 14 leave 8-20)
};

// geometry sanity: a workgroup is T*C (column kernel) / T*Q (row kernels) threads
template <int L>
constexpr bool cfg_fits() {
    using CF = CfgFor<L>;
    using CC = typename ColCfgFor<L>::type;
    return CC::T * CF::C32 <= 1024 && CC::T * CF::C64 <= 1024 && CF::Cfg::T * CF::Q32 <= 1024 && CF::Cfg::T * CF::Q64 <= 1024;
}
static_assert(cfg_fits<64>() && cfg_fits<192>() && cfg_fits<256>() && cfg_fits<576>() && cfg_fits<1152>() &&
                  cfg_fits<2304>() && cfg_fits<4608>(),
              "workgroup size above 1024 threads");

}  // namespace rl
