// The priority queue of one search: CPython's heapq (queue.PriorityQueue, search.py:260-271) restated on a blocked
// memory layout.  Included by tsg_search.hip (device) and by the host harness of tests/test_host.py (g++), which
// checks pop order and queue contents against heapq itself.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TSG_HD __device__ __forceinline__
#define TSG_CLZ(x) __clz(x)
#else
#define TSG_HD static inline
#define TSG_CLZ(x) __builtin_clz((unsigned)(x))
struct uint4 {
  unsigned x, y, z, w;
};
#ifndef TSG_HEAP_BLOCKED
#define TSG_HEAP_BLOCKED 4095
#define TSG_HEAP_LIN_BASE (8 * (585 + 2) + 8)
#endif
#ifndef ND_WORDS
enum { ND_LP = 2, ND_WORDS = 8 };
#endif
#define TSG_STAMP 0
#endif

TSG_HD float tsg_u2f(uint32_t u) {
  union {
    uint32_t u;
    float f;
  } c;
  c.u = u;
  return c.f;
}
TSG_HD uint32_t tsg_f2u(float f) {
  union {
    uint32_t u;
    float f;
  } c;
  c.f = f;
  return c.u;
}

struct HeapEnt {
  float prio;
  int node;
};

TSG_HD HeapEnt heap_ld(const uint64_t* h, int i) {
  const uint64_t v = h[i];
  HeapEnt e;
  e.prio = tsg_u2f((uint32_t)v);
  e.node = (int)(v >> 32);
  return e;
}
TSG_HD void heap_st(uint64_t* h, int i, HeapEnt e) {
  h[i] = (uint64_t)tsg_f2u(e.prio) | ((uint64_t)(uint32_t)e.node << 32);
}
// (priority, SearchNode) tuple order of queue.PriorityQueue: priority, then SearchNode.__lt__ on the
// node's CURRENT log_prob (search.py:48-52); the same node never compares less than itself.
TSG_HD bool ent_less(HeapEnt a, HeapEnt b, const int32_t* nodes) {
  if (a.prio != b.prio) return a.prio < b.prio;
  if (a.node == b.node) return false;
  return tsg_u2f((uint32_t)nodes[(size_t)a.node * ND_WORDS + ND_LP]) <
         tsg_u2f((uint32_t)nodes[(size_t)b.node * ND_WORDS + ND_LP]);
}
TSG_HD HeapEnt ent_of(uint64_t v) {
  HeapEnt e;
  e.prio = tsg_u2f((uint32_t)v);
  e.node = (int)(v >> 32);
  return e;
}
// ---- the queue in memory: a blocked heap.  The logical structure is CPython's binary heap (entry j, 1-based, has
// the children 2j and 2j+1; every comparison and every move below is heapq's, in heapq's order); only the place of
// entry j in memory differs.  Three levels of the tree share a 64-byte block (local root at word 1, its children
// at words 2-3, their children at words 4-7; word 0 unused), blocks are numbered like an 8-ary heap, and the
// stretch starts one block late, so that the two blocks rooted at the children of an entry form one aligned
// 128-byte line.  A sift then costs one cache line per three levels instead of one per level and sibling pair:
// what a launch of this kernel costs follows the number of distinct lines its lanes ask for.
// Entries beyond the first twelve levels (j >= 4096: queues that long are rare, and blocks that deep would be
// nearly empty) lie behind the blocks in plain heap order.
TSG_HD int heap_addr(int j) {  // word offset of entry j >= 1 in the slot's stretch
  if (j > TSG_HEAP_BLOCKED) return TSG_HEAP_LIN_BASE + (j - TSG_HEAP_BLOCKED - 1);
  const int d = 31 - TSG_CLZ(j);
  const int L = (d * 11) >> 5;  // d / 3
  const int r = d - 3 * L;
  const int p8 = 1 << (3 * L);
  const int pre = 0x49249249 & (p8 - 1);  // (8^L - 1) / 7 blocks above block level L
  return ((pre + (j >> r) - p8 + 1) << 3) + (1 << r) + (j & ((1 << r) - 1));
}

// heapq.heappush == append + _siftdown(heap, 0, len-1).  The ancestors of the insertion point have known
// addresses, so TSG_PUSH_ANC of them are fetched per round trip instead of one (they lie in two blocks at most
// for three of them).  A push climbs 0.35 levels on average.
#ifndef TSG_PUSH_ANC
#define TSG_PUSH_ANC 3
#endif
TSG_HD void heap_push(uint64_t* h, int& n, HeapEnt item, const int32_t* nodes, int* moves = nullptr) {
  int j = ++n;  // 1-based index of the new entry
  while (j > 1) {
    uint64_t av[TSG_PUSH_ANC];
    const int j0 = j;
#pragma unroll
    for (int k = 0; k < TSG_PUSH_ANC; ++k) {
      const int a = j0 >> (k + 1);
      av[k] = h[heap_addr(a > 0 ? a : 1)];
    }
    bool stop = false;
#pragma unroll
    for (int k = 0; k < TSG_PUSH_ANC; ++k) {
      if (stop || j == 1) continue;
      const HeapEnt par = ent_of(av[k]);
      if (!ent_less(item, par, nodes)) {
        stop = true;
      } else {
        h[heap_addr(j)] = av[k];
        j = j0 >> (k + 1);
        if (moves) ++*moves;
      }
    }
    if (stop) break;
  }
  heap_st(h, heap_addr(j), item);
}

// heapq.heappop == pop last, put it at the root, _siftup (bubble the smaller child up to a leaf, then
// _siftdown).  The root's block gives the first two levels; from there on one aligned 128-byte line (the two
// blocks rooted at the children of the current entry) gives three levels per round trip; the decisions are the
// same comparisons in the same order.  _siftdown's first comparison is against the entry the sift moved last,
// which is still in a register.
// `side`: the caller's own chain of look-ups for the entry that is being popped (its node record, then what the
// record addresses, then what that addresses) rides along with the sift: side.begin(node) is called as soon as the
// popped entry is known (one round trip into the pop), side.step() once per round trip of the sift, right after
// that round trip's loads have been requested; the sift only reads node records and writes the queue, so the two
// chains are independent and share their round trips instead of following each other.  The caller finishes what is
// left of its chain after the pop (side.finish()).  NoSide: nothing rides along (host harness, tests).
struct HeapNoSide {
  TSG_HD void begin(int) {}
  TSG_HD void step() {}
};
template <class Side>
TSG_HD HeapEnt heap_pop(uint64_t* h, int& n, const int32_t* nodes, Side& side) {
  const HeapEnt last = ent_of(h[heap_addr(n)]);
  --n;
  if (n == 0) {
    side.begin(last.node);
    return last;
  }
  const uint4* rb = reinterpret_cast<const uint4*>(h + 8);  // the root's block: words 8..15
  const uint4 r0 = rb[0], r1 = rb[1], r2 = rb[2], r3 = rb[3];
  auto w64 = [](unsigned lo, unsigned hi) { return (uint64_t)lo | ((uint64_t)hi << 32); };
  const HeapEnt ret = ent_of(w64(r0.z, r0.w));
  side.begin(ret.node);
  int j = 1;
  uint64_t moved = 0;  // the entry that went from `j` to its parent in the latest step
  if (2 <= n) {
    const uint64_t c0 = w64(r1.x, r1.y), c1 = w64(r1.z, r1.w);
    int s1 = 0;
    if (3 <= n && !ent_less(ent_of(c0), ent_of(c1), nodes)) s1 = 1;
    moved = s1 ? c1 : c0;
    h[8 + 1] = moved;
    j = 2 + s1;
    if (2 * j <= n) {
      const uint64_t gl = s1 ? w64(r3.x, r3.y) : w64(r2.x, r2.y), gr = s1 ? w64(r3.z, r3.w) : w64(r2.z, r2.w);
      int s2 = 0;
      if (2 * j + 1 <= n && !ent_less(ent_of(gl), ent_of(gr), nodes)) s2 = 1;
      moved = s2 ? gr : gl;
      h[8 + j] = moved;
      j = 2 * j + s2;
    }
  }
  // j is the last level of its block: its children are the local roots of a pair of blocks
  while (2 * j <= n && 2 * j <= TSG_HEAP_BLOCKED) {
    int c = 2 * j;
    const int pair = heap_addr(c) - 1;  // word 0 of the left child's block (128-byte aligned)
    const uint4* pb = reinterpret_cast<const uint4*>(h + pair);
    uint4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = pb[i];
    side.step();
    // level 1: the two local roots
    const uint64_t l1 = w64(q[0].z, q[0].w), x1 = w64(q[4].z, q[4].w);
    int s1 = 0;
    if (c + 1 <= n && !ent_less(ent_of(l1), ent_of(x1), nodes)) s1 = 1;
    moved = s1 ? x1 : l1;
    h[heap_addr(j)] = moved;
    j = c + s1;
    const int blk = pair + 8 * s1;
    const uint4 b1 = s1 ? q[5] : q[1], b2 = s1 ? q[6] : q[2], b3 = s1 ? q[7] : q[3];
    c = 2 * j;
    if (c > n) break;
    // level 2: words 2, 3 of the chosen block
    const uint64_t l2 = w64(b1.x, b1.y), x2 = w64(b1.z, b1.w);
    int s2 = 0;
    if (c + 1 <= n && !ent_less(ent_of(l2), ent_of(x2), nodes)) s2 = 1;
    moved = s2 ? x2 : l2;
    h[blk + 1] = moved;
    j = c + s2;
    c = 2 * j;
    if (c > n) break;
    // level 3: words 4, 5 or 6, 7
    const uint64_t l3 = s2 ? w64(b3.x, b3.y) : w64(b2.x, b2.y), x3 = s2 ? w64(b3.z, b3.w) : w64(b2.z, b2.w);
    int s3 = 0;
    if (c + 1 <= n && !ent_less(ent_of(l3), ent_of(x3), nodes)) s3 = 1;
    moved = s3 ? x3 : l3;
    h[blk + 2 + s2] = moved;
    j = c + s3;
  }
  while (2 * j <= n) {  // below the blocks: one level per round trip, the two children side by side
    const int c = 2 * j;
    const uint64_t l = h[heap_addr(c)], x = h[heap_addr(c + 1 <= n ? c + 1 : c)];
    side.step();
    int s = 0;
    if (c + 1 <= n && !ent_less(ent_of(l), ent_of(x), nodes)) s = 1;
    moved = s ? x : l;
    h[heap_addr(j)] = moved;
    j = c + s;
  }
  // _siftdown(heap, 0, j): the parent of `j` now holds `moved`
  if (j > 1) {
    if (ent_less(last, ent_of(moved), nodes)) {
      h[heap_addr(j)] = moved;
      j >>= 1;
      while (j > 1) {
        const int pp = j >> 1;
        const uint64_t par = h[heap_addr(pp)];
        if (!ent_less(last, ent_of(par), nodes)) break;
        h[heap_addr(j)] = par;
        j = pp;
      }
    }
  }
  heap_st(h, heap_addr(j), last);
  return ret;
}
TSG_HD HeapEnt heap_pop(uint64_t* h, int& n, const int32_t* nodes) {
  HeapNoSide none;
  return heap_pop(h, n, nodes, none);
}

