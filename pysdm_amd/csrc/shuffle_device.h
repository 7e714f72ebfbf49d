// shuffle_device.h -- packed event records of the binned shuffle build and the backward walk
#pragma once
#include "common.h"

// Record of one position: own target j (-1: none), the first hitting events inline (-1: none),
// initial content of the position (its super-droplet id) and a flag "further hits in the
// overflow list".  Four layouts:
//   PackRec     16 B: j, 2 inline hits, id | flag in bit 31                      (any size)
//   PackRec21   16 B: 21-bit fields - j, 4 inline hits, id, flag    (positions and ids < 2^21 - 1)
//   PackRec24   16 B: 24-bit fields - j, 3 inline hits, id, flag    (positions and ids < 2^24 - 1)
// A position is hit by Poisson(1) events: 8 % of them overflow two inline slots, 1.9 % three,
// 0.4 % four.  An
// overflow costs a walk two or more *dependent* loads (list head, links) on top of the record's,
// and a wavefront walks 128 positions per round - with two slots nearly every round of every
// wavefront waited for such a chain (k_pair_all 74 us, 53 us with the lists ignored); hence the
// packed layout wherever it fits.
struct __align__(16) PackRec { int32_t j, s0, s1, val; };
struct __align__(16) PackRec21 { uint64_t lo, hi; };
struct __align__(16) PackRec24 { uint64_t lo, hi; };
#define P21_NONE 0x1FFFFFu
#define P21_MAX 0x1FFFFE  // largest count of positions / ids the packed layout holds

#define P24_NONE 0xFFFFFFu
#define P24_MAX 0xFFFFFE
#define SDM_REC_PLAIN 0
#define SDM_REC_P21 1
#define SDM_REC_P24 2
// SDM_REC_CHAIN (round 4): no records at all - the walk's whole case analysis is done by the build,
// which sees every event that touches a position together in LDS, and handed over as successor
// words.  Let F(q, t) be the content of position q once all events with index > t have been applied
// (the chain runs from the top index down: index_methods.py:32-43); the final content of p is
// F(p, lo).  With e = the smallest event index > t that touches q (its own event q, or an event
// that hits it): F(q, t) = idx0[q] if there is none, F(j_q, q) if e is q's own event, F(e, e) if
// e hits q.  Only two families of arguments ever occur: T(i) = F(i, i) - the content of position
// i just before its own event - and S(i) = F(j_i, i) - the content of i's target just before i's
// event - and each is the other family's value further up or a terminal id:
//   first[p] = what F(p, lo) is: T(e) / S(p) / the id              (by p's bin, coalesced)
//   tsucc[i] = what T(i) is: T(next hit on i above i) or idx0[i]   (by i's bin, coalesced)
//   ssucc[l] = what S(i) is: S(j_i) / T(next hit on j_i above i) / idx0[j_i]
//              (by j_i's bin: it holds all of j_i's events) - stored where that bin READ event i:
//              at i's place l in the tile-sorted event array, so the write is as coalesced as
//              the read was (runs of ~16 events per tile and bin).  By event index it would be
//              2^20 scattered 4-byte stores: measured, 14 us on top of a 14-us build
//              (profiles/r04_chain_formats.json).  Every word that names an S carries
//              that place (the tile sort records it per event: `loc`).
// A word is {kind: 0 = super-droplet id, 1 = T, 2 = S} << 30 | payload (T: event index, S: place
// in the sorted array).  The walk is then a chain of single 4-byte look-ups (no decoding, no
// overflow lists, nothing to compare) in two 4-byte tables - 8 MB at 2^20 positions where the
// records took 16 - and its first look-up, which the records needed a random read for, comes with
// the coalesced read of first[p].
#define SDM_REC_CHAIN 3
#define CHAIN_MAX 0x3FFFFFFF  // ids and positions below 2^30
#define CHAIN_T 1u
#define CHAIN_S 2u

struct ShuffleViews {
  const void *rec;  // array of the record type `fmt` names
  int fmt;
  const int32_t *ovf_head, *ovf_next;
};

#ifdef __HIPCC__
// lo: j [0,21) s0 [21,42) s1 [42,63) flag [63]; hi: s2 [0,21) s3 [21,42) id [42,63)
__device__ __forceinline__ void p21_pack(uint64_t &lo, uint64_t &hi, int32_t j, int32_t s0,
                                         int32_t s1, int32_t s2, int32_t s3, int32_t id,
                                         bool more) {
  lo = ((uint64_t)((uint32_t)j & P21_NONE)) | ((uint64_t)((uint32_t)s0 & P21_NONE) << 21) |
       ((uint64_t)((uint32_t)s1 & P21_NONE) << 42) | ((uint64_t)(more ? 1 : 0) << 63);
  hi = ((uint64_t)((uint32_t)s2 & P21_NONE)) | ((uint64_t)((uint32_t)s3 & P21_NONE) << 21) |
       ((uint64_t)(uint32_t)id << 42);
}
__device__ __forceinline__ int32_t p21_field(uint64_t w, int shift) {
  const uint32_t f = (uint32_t)(w >> shift) & P21_NONE;
  return f == P21_NONE ? -1 : (int32_t)f;
}

// lo: j [0,24) s0 [24,48) id bits 0-14 [48,63) flag [63]; hi: s1 [0,24) s2 [24,48) id bits 15-23 [48,57)
__device__ __forceinline__ void p24_pack(uint64_t &lo, uint64_t &hi, int32_t j, int32_t s0,
                                         int32_t s1, int32_t s2, int32_t id, bool more) {
  lo = ((uint64_t)((uint32_t)j & P24_NONE)) | ((uint64_t)((uint32_t)s0 & P24_NONE) << 24) |
       ((uint64_t)((uint32_t)id & 0x7FFFu) << 48) | ((uint64_t)(more ? 1 : 0) << 63);
  hi = ((uint64_t)((uint32_t)s1 & P24_NONE)) | ((uint64_t)((uint32_t)s2 & P24_NONE) << 24) |
       ((uint64_t)((uint32_t)id >> 15) << 48);
}
__device__ __forceinline__ int32_t p24_field(uint64_t w, int shift) {
  const uint32_t f = (uint32_t)(w >> shift) & P24_NONE;
  return f == P24_NONE ? -1 : (int32_t)f;
}

// uniform view of a record for the walk: own target, up to four inline hits, overflow flag, id
struct RecView { int32_t j, h0, h1, h2, h3; bool more; };
__device__ __forceinline__ RecView rec_view(const PackRec &r) {
  return {r.j, r.s0, r.s1, -1, -1, r.val < 0};
}
__device__ __forceinline__ RecView rec_view(const PackRec21 &r) {
  return {p21_field(r.lo, 0), p21_field(r.lo, 21), p21_field(r.lo, 42), p21_field(r.hi, 0),
          p21_field(r.hi, 21), (r.lo >> 63) != 0};
}
__device__ __forceinline__ RecView rec_view(const PackRec24 &r) {
  return {p24_field(r.lo, 0), p24_field(r.lo, 24), p24_field(r.hi, 0), p24_field(r.hi, 24), -1,
          (r.lo >> 63) != 0};
}
__device__ __forceinline__ int64_t rec_id(const PackRec24 &r) {
  return (int64_t)((r.lo >> 48) & 0x7FFFu) | (int64_t)(((r.hi >> 48) & 0x1FFu) << 15);
}
__device__ __forceinline__ int64_t rec_id(const PackRec &r) { return r.val & 0x7fffffff; }
__device__ __forceinline__ int64_t rec_id(const PackRec21 &r) { return (int64_t)(r.hi >> 42); }

// the event that moved the content of position q last before event e (index.hip): smallest
// event index > e among q's own event and the events that hit q; INT32_MAX if none
__device__ __forceinline__ int32_t walk_next(const RecView &v, int32_t q, int32_t e,
                                             const int32_t *__restrict__ ovf_head,
                                             const int32_t *__restrict__ ovf_next) {
  int32_t best = INT32_MAX;
  if (q > e && v.j >= 0) best = q;
  if (v.h0 > e && v.h0 < best) best = v.h0;
  if (v.h1 > e && v.h1 < best) best = v.h1;
  if (v.h2 > e && v.h2 < best) best = v.h2;
  if (v.h3 > e && v.h3 < best) best = v.h3;
  if (v.more)
    for (int32_t t = ovf_head[q]; t >= 0; t = ovf_next[t])
      if (t > e && t < best) best = t;
  return best;
}

// the record at which the walk from position p ends: it names the content of p after the whole
// swap chain of its cell (cell starts at position `lo`)
template <class REC>
__device__ __forceinline__ REC walk_packed(const REC *__restrict__ rec,
                                           const int32_t *__restrict__ ovf_head,
                                           const int32_t *__restrict__ ovf_next, int32_t p,
                                           int32_t lo) {
  int32_t e = lo, q = p;  // only events with index > e are still "in the past" of the walk
  REC r;
  for (;;) {
    r = rec[q];
    const RecView v = rec_view(r);
    const int32_t best = walk_next(v, q, e, ovf_head, ovf_next);
    if (best == INT32_MAX) break;
    // event `best` exchanged positions (best, j_best); q is one end, continue at the other
    q = (best == q) ? v.j : best;
    e = best;
  }
  return r;
}

// two walks advanced in lockstep: both record loads of a round are in flight together
template <class REC>
__device__ __forceinline__ void walk_packed2(const REC *__restrict__ rec,
                                             const int32_t *__restrict__ ovf_head,
                                             const int32_t *__restrict__ ovf_next, int32_t p0,
                                             int32_t p1, int32_t lo, REC &f0, REC &f1) {
  int32_t e0 = lo, q0 = p0, e1 = lo, q1 = p1;
  bool done0 = false, done1 = false;
  REC r0 = rec[q0], r1 = rec[q1];
  for (;;) {
    if (!done0) {
      const RecView v = rec_view(r0);
      const int32_t best = walk_next(v, q0, e0, ovf_head, ovf_next);
      if (best == INT32_MAX) { done0 = true; } else { q0 = (best == q0) ? v.j : best; e0 = best; }
    }
    if (!done1) {
      const RecView v = rec_view(r1);
      const int32_t best = walk_next(v, q1, e1, ovf_head, ovf_next);
      if (best == INT32_MAX) { done1 = true; } else { q1 = (best == q1) ? v.j : best; e1 = best; }
    }
    if (done0 && done1) break;
    // next round: both loads issued back to back
    REC n0 = r0, n1 = r1;
    if (!done0) n0 = rec[q0];
    if (!done1) n1 = rec[q1];
    r0 = n0;
    r1 = n1;
  }
  f0 = r0;
  f1 = r1;
}
__device__ __forceinline__ uint32_t chain_word(uint32_t kind, int32_t payload) {
  return (kind << 30) | (uint32_t)payload;
}
// SDM_REC_CHAIN: two walks in lockstep (rec = first, ovf_head = tsucc, ovf_next = ssucc)
__device__ __forceinline__ void walk_chain2(const uint32_t *__restrict__ first,
                                            const uint32_t *__restrict__ tsucc,
                                            const uint32_t *__restrict__ ssucc, int32_t p0,
                                            int32_t p1, int64_t &id0, int64_t &id1) {
  uint32_t n0, n1;
  if (p1 == p0 + 1 && (p0 & 1) == 0) {  // a pair slot: one 8-byte load
    const uint2 w = *(const uint2 *)(first + p0);
    n0 = w.x;
    n1 = w.y;
  } else {
    n0 = first[p0];
    n1 = first[p1];
  }
  while ((n0 | n1) >> 30) {  // both look-ups of a round are in flight together
    const uint32_t k0 = n0 >> 30, k1 = n1 >> 30;
    uint32_t m0 = n0, m1 = n1;
    if (k0) m0 = (k0 == CHAIN_T ? tsucc : ssucc)[n0 & CHAIN_MAX];
    if (k1) m1 = (k1 == CHAIN_T ? tsucc : ssucc)[n1 & CHAIN_MAX];
    n0 = m0;
    n1 = m1;
  }
  id0 = n0;
  id1 = n1;
}
__device__ __forceinline__ int64_t walk_chain(const uint32_t *__restrict__ first,
                                              const uint32_t *__restrict__ tsucc,
                                              const uint32_t *__restrict__ ssucc, int32_t p) {
  uint32_t n = first[p];
  while (n >> 30) n = ((n >> 30) == CHAIN_T ? tsucc : ssucc)[n & CHAIN_MAX];
  return n;
}

// the walks' results as super-droplet ids, for records of layout `fmt`
__device__ __forceinline__ void walk_ids2(const void *rec, int fmt, const int32_t *ovf_head,
                                          const int32_t *ovf_next, int32_t p0, int32_t p1,
                                          int32_t lo, int64_t &id0, int64_t &id1) {
  if (fmt == SDM_REC_CHAIN) {
    walk_chain2((const uint32_t *)rec, (const uint32_t *)ovf_head, (const uint32_t *)ovf_next, p0,
                p1, id0, id1);
  } else if (fmt == SDM_REC_P21) {
    PackRec21 f0, f1;
    walk_packed2((const PackRec21 *)rec, ovf_head, ovf_next, p0, p1, lo, f0, f1);
    id0 = rec_id(f0); id1 = rec_id(f1);
  } else if (fmt == SDM_REC_P24) {
    PackRec24 f0, f1;
    walk_packed2((const PackRec24 *)rec, ovf_head, ovf_next, p0, p1, lo, f0, f1);
    id0 = rec_id(f0); id1 = rec_id(f1);
  } else {
    PackRec f0, f1;
    walk_packed2((const PackRec *)rec, ovf_head, ovf_next, p0, p1, lo, f0, f1);
    id0 = rec_id(f0); id1 = rec_id(f1);
  }
}
__device__ __forceinline__ int64_t walk_id(const void *rec, int fmt, const int32_t *ovf_head,
                                           const int32_t *ovf_next, int32_t p, int32_t lo) {
  if (fmt == SDM_REC_CHAIN)
    return walk_chain((const uint32_t *)rec, (const uint32_t *)ovf_head,
                      (const uint32_t *)ovf_next, p);
  if (fmt == SDM_REC_P21)
    return rec_id(walk_packed((const PackRec21 *)rec, ovf_head, ovf_next, p, lo));
  if (fmt == SDM_REC_P24)
    return rec_id(walk_packed((const PackRec24 *)rec, ovf_head, ovf_next, p, lo));
  return rec_id(walk_packed((const PackRec *)rec, ovf_head, ovf_next, p, lo));
}
#endif
