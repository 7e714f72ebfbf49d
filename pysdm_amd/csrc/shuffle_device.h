// shuffle_device.h -- packed event records of the binned shuffle build and the backward walk
#pragma once
#include "common.h"

// own target j (-1: none), first / second hitting event (-1: none), initial content of the
// position (int32) with bit 31 = "further hits in the overflow list"
struct __align__(16) PackRec { int32_t j, s0, s1, val; };

struct ShuffleViews {
  const PackRec *rec;
  const int32_t *ovf_head, *ovf_next;
};

#ifdef __HIPCC__
// content of position p after the whole swap chain of its cell (cell starts at position `lo`):
// walk the swap history backwards (see index.hip)
__device__ __forceinline__ int64_t walk_packed(const PackRec *__restrict__ rec,
                                               const int32_t *__restrict__ ovf_head,
                                               const int32_t *__restrict__ ovf_next, int32_t p,
                                               int32_t lo) {
  int32_t e = lo, q = p;  // only events with index > e are still "in the past" of the walk
  PackRec r;
  for (;;) {
    r = rec[q];
    int32_t best = INT32_MAX;
    if (q > e && r.j >= 0) best = q;
    if (r.s0 > e && r.s0 < best) best = r.s0;
    if (r.s1 > e && r.s1 < best) best = r.s1;
    if (r.val < 0)
      for (int32_t t = ovf_head[q]; t >= 0; t = ovf_next[t])
        if (t > e && t < best) best = t;
    if (best == INT32_MAX) break;
    // event `best` exchanged positions (best, j_best); q is one end, continue at the other
    q = (best == q) ? r.j : best;
    e = best;
  }
  return (int64_t)(r.val & 0x7fffffff);
}

// two walks advanced in lockstep: both record loads of a round are in flight together
__device__ __forceinline__ void walk_packed2(const PackRec *__restrict__ rec,
                                             const int32_t *__restrict__ ovf_head,
                                             const int32_t *__restrict__ ovf_next, int32_t p0,
                                             int32_t p1, int32_t lo, int64_t &v0, int64_t &v1) {
  int32_t e0 = lo, q0 = p0, e1 = lo, q1 = p1;
  bool done0 = false, done1 = false;
  PackRec r0 = rec[q0], r1 = rec[q1];
  for (;;) {
    if (!done0) {
      int32_t best = INT32_MAX;
      if (q0 > e0 && r0.j >= 0) best = q0;
      if (r0.s0 > e0 && r0.s0 < best) best = r0.s0;
      if (r0.s1 > e0 && r0.s1 < best) best = r0.s1;
      if (r0.val < 0)
        for (int32_t t = ovf_head[q0]; t >= 0; t = ovf_next[t])
          if (t > e0 && t < best) best = t;
      if (best == INT32_MAX) { done0 = true; } else { q0 = (best == q0) ? r0.j : best; e0 = best; }
    }
    if (!done1) {
      int32_t best = INT32_MAX;
      if (q1 > e1 && r1.j >= 0) best = q1;
      if (r1.s0 > e1 && r1.s0 < best) best = r1.s0;
      if (r1.s1 > e1 && r1.s1 < best) best = r1.s1;
      if (r1.val < 0)
        for (int32_t t = ovf_head[q1]; t >= 0; t = ovf_next[t])
          if (t > e1 && t < best) best = t;
      if (best == INT32_MAX) { done1 = true; } else { q1 = (best == q1) ? r1.j : best; e1 = best; }
    }
    if (done0 && done1) break;
    // next round: both loads issued back to back
    PackRec n0 = r0, n1 = r1;
    if (!done0) n0 = rec[q0];
    if (!done1) n1 = rec[q1];
    r0 = n0;
    r1 = n1;
  }
  v0 = (int64_t)(r0.val & 0x7fffffff);
  v1 = (int64_t)(r1.val & 0x7fffffff);
}
#endif
