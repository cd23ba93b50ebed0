// bpsw_rescue_skip.h -- which orientations of an anchor need no mate-SW: the test both the boundary-1 host layer (bpsw_rescue.cpp:
// speculation and replay) and the JNI shim (bpsw_jni.cpp: which pairs of a mateSWJNI call it has to unmarshal at all) apply.  Plain C++,
// no HIP: the one place the predicate lives, so that the shim's pre-selection cannot drift from what the library would launch.
#pragma once
#include <stdint.h>

namespace bpsw {

// native/bwamem_pair.c:27-34 (mem_infer_dir): orientation 0..3 of a pair of hits by their starts on the doubled reference, and their distance
inline int rescue_infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t* dist) {
  const bool r1 = b1 >= l_pac, r2 = b2 >= l_pac;
  const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
  *dist = p2 > b1 ? p2 - b1 : b1 - p2;
  return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

// skip[r] = 1: orientation r of the anchor starting at a_rb needs no SW -- its statistics failed (bit r of failed_mask) or one of the
// mate's hits (starts mate_rb[0..n), read with `stride` bytes between them) already lies at a proper distance in that orientation
// (native/bwamem_pair.c:119-124 / MemSamPe.scala:1131-1146; scala_narrow: MemSamPe.scala:1137-1138 narrows the distance to Int).
inline void rescue_skip_flags(int64_t l_pac, const int32_t low[4], const int32_t high[4], int failed_mask, bool scala_narrow, int64_t a_rb,
                              const void* mate_rb, size_t stride, size_t n_mates, int skip[4]) {
  for (int r = 0; r < 4; ++r) skip[r] = (failed_mask >> r) & 1;
  const char* p = (const char*)mate_rb;
  for (size_t mi = 0; mi < n_mates; ++mi, p += stride) {
    int64_t m_rb;
    __builtin_memcpy(&m_rb, p, 8);
    int64_t dist;
    const int r = rescue_infer_dir(l_pac, a_rb, m_rb, &dist);
    if (scala_narrow) dist = (int64_t)(int32_t)dist;
    if (dist >= low[r] && dist <= high[r]) skip[r] = 1;
  }
}

}  // namespace bpsw
