// bpsw_wave.h -- wave64 cross-lane helpers (DPP) shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace bpsw {

constexpr int NEG = -(1 << 29);

// DPP controls (gfx9 encodings)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// lanes whose source is out of range or masked off keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// lane l <- lane l-1 ; lane 0 <- old
__device__ __forceinline__ int wave_shr1(int old, int src) { return dpp_mov<DPP_WAVE_SHR1, 0xf>(old, src); }

// inclusive max-scan over the 64 lanes; lane 63 ends up with the wave maximum
__device__ __forceinline__ int wave_scan_max(int v) {
  v = max(v, dpp_mov<DPP_ROW_SHR1, 0xf>(NEG, v));
  v = max(v, dpp_mov<DPP_ROW_SHR2, 0xf>(NEG, v));
  v = max(v, dpp_mov<DPP_ROW_SHR4, 0xf>(NEG, v));
  v = max(v, dpp_mov<DPP_ROW_SHR8, 0xf>(NEG, v));
  v = max(v, dpp_mov<DPP_ROW_BCAST15, 0xa>(NEG, v));
  v = max(v, dpp_mov<DPP_ROW_BCAST31, 0xc>(NEG, v));
  return v;
}
// inclusive prefix sum / prefix minimum over the 64 lanes (same DPP ladder)
__device__ __forceinline__ int wave_scan_add(int v) {
  v += dpp_mov<DPP_ROW_SHR1, 0xf>(0, v);
  v += dpp_mov<DPP_ROW_SHR2, 0xf>(0, v);
  v += dpp_mov<DPP_ROW_SHR4, 0xf>(0, v);
  v += dpp_mov<DPP_ROW_SHR8, 0xf>(0, v);
  v += dpp_mov<DPP_ROW_BCAST15, 0xa>(0, v);
  v += dpp_mov<DPP_ROW_BCAST31, 0xc>(0, v);
  return v;
}
constexpr int POS = 1 << 29;
__device__ __forceinline__ int wave_scan_min(int v) {
  v = min(v, dpp_mov<DPP_ROW_SHR1, 0xf>(POS, v));
  v = min(v, dpp_mov<DPP_ROW_SHR2, 0xf>(POS, v));
  v = min(v, dpp_mov<DPP_ROW_SHR4, 0xf>(POS, v));
  v = min(v, dpp_mov<DPP_ROW_SHR8, 0xf>(POS, v));
  v = min(v, dpp_mov<DPP_ROW_BCAST15, 0xa>(POS, v));
  v = min(v, dpp_mov<DPP_ROW_BCAST31, 0xc>(POS, v));
  return v;
}
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_scan_max(v), 63); }

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace bpsw
