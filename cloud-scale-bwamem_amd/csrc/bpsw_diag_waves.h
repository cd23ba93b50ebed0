// Diagnostics builds only (tools/build_variant.sh waves -DBPSW_DIAG_WAVES; tools/wave_placement.py): every wavefront of the
// extension and rescue kernels logs when it ran (s_memrealtime, 100 MHz) and where (HW_ID: SIMD / CU / SE, XCC_ID), so that a
// bench run can be asked how the workgroup dispatcher spreads concurrent small launches over the 256 CUs and how long the
// waves of a launch live compared with the launch.  The product build compiles none of this.
#pragma once
#ifdef BPSW_DIAG_WAVES
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

namespace bpsw {
constexpr unsigned DIAG_WAVE_CAP = 1u << 21;
// one log per translation unit (no relocatable device code in this build): BPSW_DIAG_WAVES_DEFINE(name) in the .hip file
#define BPSW_DIAG_WAVES_DEFINE(NAME)                                                                                       \
  static __device__ unsigned g_diag_wave_n;                                                                                \
  static __device__ uint4 g_diag_wave_log[2 * bpsw::DIAG_WAVE_CAP];                                                            \
  extern "C" int bpsw_diag_dump_waves_##NAME(const char* path) {                                                           \
    unsigned n = 0;                                                                                                        \
    if (hipDeviceSynchronize() != hipSuccess) return -1;                                                                   \
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_diag_wave_n), sizeof(n)) != hipSuccess) return -2;                            \
    if (n > bpsw::DIAG_WAVE_CAP) n = bpsw::DIAG_WAVE_CAP;                                                                  \
    std::vector<uint4> v(n ? 2 * (size_t)n : 1);                                                                                       \
    if (n && hipMemcpyFromSymbol(v.data(), HIP_SYMBOL(g_diag_wave_log), 2 * sizeof(uint4) * (size_t)n) != hipSuccess) return -3; \
    FILE* f = fopen(path, "wb");                                                                                           \
    if (!f) return -4;                                                                                                     \
    fwrite(v.data(), sizeof(uint4), 2 * (size_t)n, f);                                                                                 \
    fclose(f);                                                                                                             \
    return (int)n;                                                                                                         \
  }                                                                                                                        \
  __device__ __forceinline__ void diag_wave_end(const unsigned long long t0, const int kind, const void* tag_ptr, const int lane,  \
                                                const uint4 extra = make_uint4(0u, 0u, 0u, 0u)) {                           \
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();                                                        \
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);       \
    if (lane == 0) {                                                                                                       \
      const unsigned k = atomicAdd(&g_diag_wave_n, 1u);                                                                    \
      if (k < bpsw::DIAG_WAVE_CAP) {                                                                                       \
        g_diag_wave_log[2 * k] = make_uint4((unsigned)t0, (unsigned)t1, hw,                                                \
                                            (xcc & 0xfu) | ((unsigned)kind << 4) | ((unsigned)(((unsigned long long)tag_ptr) >> 8) << 8)); \
        g_diag_wave_log[2 * k + 1] = extra;                                                                                \
      }                                                                                                                    \
    }                                                                                                                      \
  }
#define BPSW_DIAG_WAVE_BEGIN() const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime()
#define BPSW_DIAG_WAVE_END(KIND, TAG, LANE) diag_wave_end(diag_t0, KIND, TAG, LANE)
// per task (ext_kernel): the wave's last task, how long it took, the tasks it swept and the longest of them (10 ns ticks)
#define BPSW_DIAG_TASKS_DECL() unsigned diag_last = 0u, diag_last_dur = 0u, diag_n = 0u, diag_max = 0u; unsigned long long diag_tt0 = 0ull
#define BPSW_DIAG_TASK_BEGIN(TASK) do { diag_last = (unsigned)(TASK); diag_tt0 = __builtin_amdgcn_s_memrealtime(); } while (0)
#define BPSW_DIAG_TASK_END() do { diag_last_dur = (unsigned)(__builtin_amdgcn_s_memrealtime() - diag_tt0); ++diag_n; diag_max = diag_last_dur > diag_max ? diag_last_dur : diag_max; } while (0)
#define BPSW_DIAG_WAVE_END_TASKS(KIND, TAG, LANE) diag_wave_end(diag_t0, KIND, TAG, LANE, make_uint4(diag_last, diag_last_dur, diag_n, diag_max))
// rescue kernel: four words of the wave's (last) job pair
#define BPSW_DIAG_DUO_DECL() unsigned diag_d0 = 0u, diag_d1 = 0u, diag_d2 = 0u, diag_d3 = 0u
#define BPSW_DIAG_DUO_SET(A, B, CC, DD) do { diag_d0 = (unsigned)(A); diag_d1 = (unsigned)(B); diag_d2 = (unsigned)(CC); diag_d3 = (unsigned)(DD); } while (0)
#define BPSW_DIAG_WAVE_END_DUO(KIND, TAG, LANE) diag_wave_end(diag_t0, KIND, TAG, LANE, make_uint4(diag_d0, diag_d1, diag_d2, diag_d3))
}  // namespace bpsw
#else
#define BPSW_DIAG_WAVES_DEFINE(NAME)
#define BPSW_DIAG_WAVE_BEGIN()
#define BPSW_DIAG_WAVE_END(KIND, TAG, LANE)
#define BPSW_DIAG_TASKS_DECL()
#define BPSW_DIAG_TASK_BEGIN(TASK)
#define BPSW_DIAG_TASK_END()
#define BPSW_DIAG_WAVE_END_TASKS(KIND, TAG, LANE)
#define BPSW_DIAG_DUO_DECL()
#define BPSW_DIAG_DUO_SET(A, B, CC, DD)
#define BPSW_DIAG_WAVE_END_DUO(KIND, TAG, LANE)
#endif
