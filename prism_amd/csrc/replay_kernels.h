// Device code of the replay ring + PER trees (kernels and the block-level routines the fused step
// kernels reuse).  Semantics restated in oracle/per_oracle.c; reference call sites:
//   /root/reference/prism/experience/timestep_buffer.py:32-54,79-238 and the torchrl
//   PrioritizedSampler it wraps (prism/factory/exp_buffer_factory.py:22-28).
//
// All of this is latency-bound integer/fp32 pointer chasing (17-21 dependent tree levels), so the
// kernels are shaped for few launches, wave-parallelism ACROSS samples, LDS-cached tree tops and
// coalesced 16-byte row copies — not for MFMA.
#pragma once
#include <float.h>

#include "common.h"

namespace prism {

constexpr int TOP_LEVELS = 8;                  // nodes [1, 2^8) of the sum tree cached in LDS (2 KB; 11 levels: +0.35 us on the front launch, A/B)
constexpr int TOP_NODES = 1 << TOP_LEVELS;

// node n of the interleaved tree: .x = sum, .y = min
__device__ __forceinline__ float2 *tree_nodes(const prism_replay_desc &rp) { return reinterpret_cast<float2 *>(rp.tree); }

template <bool MIN>
__device__ __forceinline__ float tree_op(float a, float b) {
    if (MIN) return a < b ? a : b;
    return a + b;
}

// SegmentTree::Query(0, r) restated so that all node loads are issued in parallel (one lane per
// (level, side)) and then folded by one lane in exactly the sequential order.
// Must be called by all threads of the block; needs blockDim.x >= 128; `scratch` holds 128 floats.
template <bool MIN>
__device__ float block_tree_query(const float2 *__restrict__ v, int64_t cap, int64_t tree_size, int64_t r_in,
                                  float *scratch) {
    const float ident = MIN ? FLT_MAX : 0.0f;
    if (r_in >= tree_size) return MIN ? v[1].y : v[1].x;
    const int t = threadIdx.x;
    if (t < 128) {
        const int level = t >> 1, side = t & 1;
        int64_t l = cap, r = r_in | cap;
        float val = ident;
        bool live = true;
        for (int i = 0; i < level && live; ++i) {
            if (!(l < r)) { live = false; break; }
            if (l & 1) ++l;
            if (r & 1) --r;
            l >>= 1;
            r >>= 1;
        }
        if (live && l < r) {
            if (side == 0) {
                if (l & 1) val = MIN ? v[l].y : v[l].x;
            } else {
                if (r & 1) val = MIN ? v[r - 1].y : v[r - 1].x;
            }
        }
        scratch[t] = val;
    }
    __syncthreads();
    float ret = ident;
    // identity entries fold as no-ops for min; for the sum they add +0.0f which is exact
    // (ret is never -0.0f here), so folding every slot of a live level in order equals the
    // sequential walk (levels above the root are all identity: skip them).
    const int used = 2 * (64 - __clzll((unsigned long long)cap));
    for (int i = 0; i < used && i < 128; ++i) ret = tree_op<MIN>(ret, scratch[i]);
    __syncthreads();
    return ret;
}

// The same query for sum and min at once, split so that the caller can put the node fetch in flight
// together with its other loads: thread t < 128 fetches its node (identity if none), the values go
// to scratch[t] / scratch[128 + t], and after a barrier tree_query_fold() folds them in the
// sequential order.
__device__ __forceinline__ float2 tree_query_fetch(const float2 *__restrict__ v, int64_t cap, int64_t tree_size,
                                                   int64_t r_in) {
    float2 val = make_float2(0.0f, FLT_MAX);
    const int t = threadIdx.x;
    if (t >= 128) return val;
    if (r_in >= tree_size) return t == 0 ? v[1] : val;          // whole tree: 0 + root, min(FLT_MAX, root)
    const int level = t >> 1, side = t & 1;
    int64_t l = cap, r = r_in | cap;
    bool live = true;
    for (int i = 0; i < level && live; ++i) {
        if (!(l < r)) { live = false; break; }
        if (l & 1) ++l;
        if (r & 1) --r;
        l >>= 1;
        r >>= 1;
    }
    if (live && l < r) {
        if (side == 0) {
            if (l & 1) val = v[l];
        } else {
            if (r & 1) val = v[r - 1];
        }
    }
    return val;
}
__device__ __forceinline__ float2 tree_query_fold(const float *scratch, int64_t cap) {
    // Slots of levels above the root hold the identities (+0.0f / FLT_MAX), which fold as exact no-ops,
    // so a fixed number of slots is folded: all of them are fetched as 16-byte words first (one LDS
    // round trip instead of one per slot), then walked in the sequential order.
    constexpr int N4 = (2 * (24 + 1) + 3) / 4;                   // 2 * (TREE_MAX_LEVELS + 1) slots
    (void)cap;
    const float4 *s4 = reinterpret_cast<const float4 *>(scratch), *m4 = reinterpret_cast<const float4 *>(scratch + 128);
    float4 sv[N4], mv[N4];
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        sv[i] = s4[i];
        mv[i] = m4[i];
    }
    float s = 0.0f, m = FLT_MAX;
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        s = ((s + sv[i].x) + sv[i].y);
        s = ((s + sv[i].z) + sv[i].w);
        m = m < mv[i].x ? m : mv[i].x;
        m = m < mv[i].y ? m : mv[i].y;
        m = m < mv[i].z ? m : mv[i].z;
        m = m < mv[i].w ? m : mv[i].w;
    }
    return make_float2(s, m);
}

// tree_query_fetch for ONE WAVE, lane = level, in closed form.  With l = cap = 2^L the walk's left end stays a power of two
// (l_i = cap >> i, never odd below the root), so only right ends contribute: level i adds node (r >> i) - 1 when r >> i is
// odd and still right of l_i -- the binary decomposition of [0, r_in).  Same nodes, and lane order = slot order.
__device__ __forceinline__ float2 tree_query_fetch_wave(const float2 *__restrict__ v, int64_t cap, int64_t tree_size,
                                                        int64_t r_in) {
    float2 val = make_float2(0.0f, FLT_MAX);
    const int level = threadIdx.x & 63;
    if (r_in >= tree_size) return level == 0 ? v[1] : val;          // whole tree: 0 + root, min(FLT_MAX, root)
    const int L = 63 - __clzll((unsigned long long)cap);
    const int64_t r = (r_in | cap) >> level, l = cap >> level;
    if (level < L && (r & 1) && l < r) val = v[r - 1];
    return val;
}

// The fold for ONE WAVE whose lane t holds slot t of tree_query_fetch (levels < 32): the sum walks the slots that hold
// something in slot order (an identity slot adds +0.0f, an exact no-op, so skipping it changes nothing), the minimum is
// order-free and takes the DPP reduction.
__device__ __forceinline__ float2 tree_query_fold_wave(const float2 val) {
    unsigned long long live = __ballot(val.x != 0.0f);
    float s = 0.0f;
    while (live) {
        const int i = __builtin_ctzll(live);
        live &= live - 1;
        s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val.x), i));
    }
    float m = val.y, o;
    o = dpp_move<0xB1, 0xF>(m, m); m = o < m ? o : m;
    o = dpp_move<0x4E, 0xF>(m, m); m = o < m ? o : m;
    o = dpp_move<0x124, 0xF>(m, m); m = o < m ? o : m;
    o = dpp_move<0x128, 0xF>(m, m); m = o < m ? o : m;
    o = dpp_move<0x142, 0xA>(m, m); m = o < m ? o : m;
    o = dpp_move<0x143, 0xC>(m, m); m = o < m ? o : m;
    return make_float2(s, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63)));
}

static __global__ void replay_init_kernel(prism_replay_desc rp) {
    const int64_t n = 2 * rp.tree_capacity;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rp.tree) {
        for (int64_t i = tid; i < n; i += stride) tree_nodes(rp)[i] = make_float2(0.0f, FLT_MAX);
    }
    for (int64_t i = tid; i < rp.capacity; i += stride) {
        rp.link[i] = -1;
        rp.back[i] = -1;
        rp.flags[i] = 0;
    }
    if (tid == 0) {
        rp.per_state[0] = 1.0f;
        rp.per_state[1] = 0.0f;
        rp.per_state[2] = 0.0f;
        rp.per_state[3] = 0.0f;
        rp.status[0] = 0;
    }
}

// ---- priority write + ancestor recompute, one workgroup, one leaf per thread ----------------
// Duplicates: the sequential reference loop leaves the LAST occurrence's value in the leaf, and
// every ancestor equals op(left, right) of the final children.
//
// Each thread walks its own leaf-to-root path with the running (sum, min) of its current node in
// registers.  The sibling subtree at every level is either untouched by this batch -- then its
// value is what global memory held on entry, prefetched for all levels up front -- or it lies on
// other threads' paths.  To find those without hashing or atomics the leaves are RANKED once
// (counting sort over LDS on the key (leaf, batch position)): in rank order the nodes of a level
// are non-decreasing, so the threads sharing a node form a contiguous run [lo, hi) and the
// sibling's run, if any, starts at `hi` (even node) or ends at `lo` (odd node).  Every thread
// keeps a 16-byte record {sum, min, lo|hi, leaf} at its rank; one ds_read_b128 of the neighbour
// record + one ds_write_b128 + one LDS barrier per level.  Threads of a run hold bit-identical
// state, so any member's record serves.  Global memory is read once and written fire-and-forget.
// (VALU cost note: a wave64 instruction occupies its SIMD for 4 cycles, so the O(n^2) ranking is
// kept to one compare + one add-with-carry per key pair.)
#ifndef TREE_STAMP
#define TREE_STAMP(k)
#endif
constexpr int TREE_MAX_LEVELS = 24;             // tree_capacity <= 2^24 (checked in prism_replay_init)
constexpr int UPD_MAX = 512;                    // leaves per pass of the single-workgroup writer
constexpr int UPD_POS_BITS = 9;
constexpr int TREE_WRITE_LDS_BYTES = UPD_MAX * 8 + UPD_MAX * 4 + 2 * UPD_MAX * 16;
// LDS of the writer (bytes).  Phases: RANK = keys are written and counted into the partials; SORT = every thread sums its
// partials (others may still be summing theirs) and scatters leaf / value to its rank, then reads its run; WALK = records of
// the level walk and, in the dense form, the 256-node level.  A workgroup barrier separates RANK from SORT's scatter only
// for the keys (block_rank's second barrier), SORT from WALK for everything ("ranking scratch is dead").
constexpr unsigned TW_RANK = 1u, TW_SORT = 2u, TW_WALK = 4u;
constexpr LdsRegion TW_SORTED{0, UPD_MAX * 4, TW_SORT};                     // int32 [UPD_MAX] leaves in rank order
constexpr LdsRegion TW_VAL{UPD_MAX * 4, UPD_MAX * 4, TW_SORT};              // float [UPD_MAX] values in rank order
constexpr LdsRegion TW_PART{UPD_MAX * 8, 4 * UPD_MAX * 2, TW_RANK | TW_SORT};   // uint16 [4][UPD_MAX] ranking partials
constexpr LdsRegion TW_KEYS{UPD_MAX * 16, UPD_MAX * 8, TW_RANK};            // uint32 / uint64 [UPD_MAX] ranking keys
constexpr LdsRegion TW_REC{UPD_MAX * 12, 2 * UPD_MAX * 16, TW_WALK};        // int4 [2][UPD_MAX] level records
constexpr LdsRegion TW_DENSE{0, 256 * 8, TW_WALK};                          // float2 [256] the dense level's old values
constexpr LdsRegion TW_REGIONS[] = {TW_SORTED, TW_VAL, TW_PART, TW_KEYS, TW_REC, TW_DENSE};
static_assert(lds_layout_ok(TW_REGIONS, TREE_WRITE_LDS_BYTES), "tree writer: LDS regions live at the same time overlap");
static_assert(TW_KEYS.off % 16 == 0 && TW_REC.off % 16 == 0 && TW_DENSE.off % 16 == 0, "16-byte accessed regions");

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// #keys < me over the slice [j0, j1) of 16-byte groups
__device__ __forceinline__ uint32_t count_less(const uint32_t *keys, int j0, int j1, uint32_t me) {
    const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
    uint32_t n = 0;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const uint4 k = k4[j];
        n += (uint32_t)(k.x < me) + (uint32_t)(k.y < me) + (uint32_t)(k.z < me) + (uint32_t)(k.w < me);
    }
    return n;
}
__device__ __forceinline__ uint32_t count_less(const uint64_t *keys, int j0, int j1, uint64_t me) {
    const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(keys);
    uint32_t n = 0;
#pragma unroll 4
    for (int j = 2 * j0; j < 2 * j1; ++j) {
        const ulonglong2 k = k2[j];
        n += (uint32_t)(k.x < me) + (uint32_t)(k.y < me);
    }
    return n;
}

// rank of (leaf, tid) among the cnt keys; K = uint32_t while leaf << UPD_POS_BITS fits, else uint64_t
template <typename K>
__device__ __forceinline__ int block_rank(int32_t my_idx, int cnt, K *s_key, uint16_t *s_part, int bd) {
    const int tid = threadIdx.x;
    const int cnt4 = (cnt + 3) & ~3;
    if (tid < cnt4) s_key[tid] = tid < cnt ? (((K)(uint32_t)my_idx << UPD_POS_BITS) | (K)tid) : ~(K)0;
    lds_only_barrier();
    int cntp = 64;
    while (cntp < cnt) cntp <<= 1;
    int parts = bd / cntp;
    parts = parts > 4 ? 4 : (parts < 1 ? 1 : parts);
    const int e = tid & (cntp - 1), part = tid / cntp;
    if (part < parts && e < cnt) {
        const int n4 = cnt4 >> 2;
        s_part[part * UPD_MAX + e] = (uint16_t)count_less(s_key, part * n4 / parts, (part + 1) * n4 / parts, s_key[e]);
    }
    lds_only_barrier();
    int rank = 0;
    if (tid < cnt)
        for (int q = 0; q < parts; ++q) rank += s_part[q * UPD_MAX + tid];
    return rank;
}

// The writer comes in three pieces so that a caller may run them in different kernels:
//   tree_sib_prefetch   sibling values of a leaf's path for all levels, into registers
//   tree_write_prepare  ranking, runs of equal leaves, winning value -> TreePrep
//   tree_write_levels   leaf write + level-by-level ancestor recompute
// Thread t < cnt carries one leaf; cnt <= min(UPD_MAX, blockDim.x); `lds`: TREE_WRITE_LDS_BYTES.
struct TreePrep {
    int rank, lo, hi;      // position in (leaf, batch position) order; run of equal leaves [lo, hi)
    float val;             // value of the last occurrence of this leaf
};
template <int NL = TREE_MAX_LEVELS>      // NL: levels the walk may cover (fewer when the top of the tree is recomputed whole)
struct SibRegs {
    float s[NL], m[NL];
};

// `sib` (optional): sib[s * sib_stride + t] = {sum, min} of the sibling of thread t's path node at
// level s as of entry, recorded by whoever walked those paths last (the sampling descent reads
// both children of every path node anyway); NULL -> read them from the tree.
template <int NL>
__device__ __forceinline__ void tree_sib_prefetch(const prism_replay_desc &rp, int32_t leaf, bool active, int levels,
                                                  const float2 *__restrict__ sib, int sib_stride, SibRegs<NL> &r) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < NL; ++s) {
        r.s[s] = 0.f;
        r.m[s] = 0.f;
        if (s < levels && active) {
            const float2 v = sib ? sib[s * sib_stride + tid] : tree_nodes(rp)[(leaf >> s) ^ 1];
            r.s[s] = v.x;
            r.m[s] = v.y;
        }
    }
}

__device__ __forceinline__ TreePrep tree_write_prepare(int32_t my_idx, int32_t leaf, float my_val, int cnt, int levels,
                                                       char *lds, int bd) {
    const int tid = threadIdx.x;
    const bool active = tid < cnt;
    int32_t *s_sorted = reinterpret_cast<int32_t *>(lds + TW_SORTED.off);
    float *s_val = reinterpret_cast<float *>(lds + TW_VAL.off);
    uint16_t *s_part = reinterpret_cast<uint16_t *>(lds + TW_PART.off);
    // ranking keys alias the records, BEHIND the partials: with three or four thread groups counting (1024 threads on
    // 256 leaves or fewer) groups 2 and 3 store their counts at bytes 6144.., where the keys used to start -- while other
    // waves were still reading them (a rare wrong rank: the one-in-fifty mismatch of the fused-vs-unfused test; with the
    // keys at byte 6144 the static_assert on TW_REGIONS fails)
    char *s_keys = lds + TW_KEYS.off;
    TreePrep p;
    p.rank = levels + UPD_POS_BITS <= 32 ? block_rank<uint32_t>(my_idx, cnt, reinterpret_cast<uint32_t *>(s_keys), s_part, bd)
                                         : block_rank<uint64_t>(my_idx, cnt, reinterpret_cast<uint64_t *>(s_keys), s_part, bd);
    TREE_STAMP(25);
    if (active) {
        s_sorted[p.rank] = leaf;
        s_val[p.rank] = my_val;
    }
    lds_only_barrier();
    TREE_STAMP(26);
    // my run of equal leaves (almost always just me); its last member is the last occurrence
    p.lo = p.rank;
    p.hi = p.rank + 1;
    p.val = 0.f;
    if (active) {
        while (p.lo > 0 && s_sorted[p.lo - 1] == leaf) --p.lo;
        while (p.hi < cnt && s_sorted[p.hi] == leaf) ++p.hi;
        p.val = s_val[p.hi - 1];
    }
    lds_only_barrier();                                         // ranking scratch is dead from here
    return p;
}

template <int NL>
__device__ __forceinline__ void tree_write_levels(const prism_replay_desc &rp, int32_t leaf, const TreePrep &p, int cnt,
                                                  int levels, char *lds, const SibRegs<NL> &sr) {
    const int tid = threadIdx.x;
    const bool active = tid < cnt;
    int4 *s_rec = reinterpret_cast<int4 *>(lds + TW_REC.off);           // [2][UPD_MAX]
    const int rank = p.rank;
    int lo = p.lo, hi = p.hi;
    float cs = p.val, cm = p.val;
    if (active) {
        s_rec[rank] = make_int4(__float_as_int(cs), __float_as_int(cm), lo | (hi << 16), leaf);
        tree_nodes(rp)[leaf] = make_float2(cs, cm);
    }
    // Every prefetched sibling value lands HERE, once, and is then laundered through an empty asm: the
    // compiler otherwise keeps "a load may still be pending" alive through the level bodies and puts an
    // s_waitcnt vmcnt(0) into each of them -- which also waits for that level's fire-and-forget store.
    float ss[NL], sm[NL];
#pragma unroll
    for (int s = 0; s < NL; ++s) {
        ss[s] = sr.s[s];
        sm[s] = sr.m[s];
        asm volatile("" : "+v"(ss[s]), "+v"(sm[s]));
    }
    lds_only_barrier();
    TREE_STAMP(5);
#pragma unroll
    for (int s = 0; s < NL; ++s) {
        if (s < levels) {                                       // uniform
            const int4 *buf = s_rec + (s & 1) * UPD_MAX;
            int4 *nbuf = s_rec + ((s + 1) & 1) * UPD_MAX;
            if (active) {
                const int32_t child = leaf >> s;
                const bool odd = child & 1;
                // branch-free: ONE 16-byte read of the neighbouring record (my own when there is none)
                const bool has = odd ? lo > 0 : hi < cnt;
                const int4 q = buf[has ? (odd ? lo - 1 : hi) : rank];
                const bool meet = has && (q.w >> s) == (child ^ 1);      // the neighbouring run is my sibling
                const float os = meet ? __int_as_float(q.x) : ss[s], om = meet ? __int_as_float(q.y) : sm[s];
                lo = meet ? min(lo, q.z & 0xffff) : lo;
                hi = meet ? max(hi, q.z >> 16) : hi;
                const float ls = odd ? os : cs, rs = odd ? cs : os;
                const float lm = odd ? om : cm, rm = odd ? cm : om;
                cs = ls + rs;
                cm = lm < rm ? lm : rm;
                tree_nodes(rp)[leaf >> (s + 1)] = make_float2(cs, cm);
                nbuf[rank] = make_int4(__float_as_int(cs), __float_as_int(cm), lo | (hi << 16), leaf);
            }
            lds_only_barrier();
            TREE_STAMP(6 + s);
        }
    }
}

// ---- the top of the tree, recomputed whole ------------------------------------------------------------------------
// A caller whose batch fits one pass may stop the run-merging walk TREE_DENSE_LEVELS below the root -- at the level that
// has 256 nodes -- by handing `levels - TREE_DENSE_LEVELS` to the routines above, and let ONE wave recompute everything
// above: the 256 nodes (old values fetched by tree_dense_fetch, the touched ones replaced from the walk's
// last records) are folded pairwise in registers, lane to lane, and all 255 nodes above them stored.  A parent is
// left + right and min(left, right) whichever way it is reached, so the tree comes out bit for bit as from the
// level-by-level walk -- which pays a workgroup barrier and an LDS round trip for each of those eight levels (15 k ticks
// for 17 levels, stand-alone).  Needs levels >= TREE_DENSE_LEVELS and `nthr` >= 64 threads still running.
constexpr int TREE_DENSE_LEVELS = 8;
// old values of the 256-node level -> LDS without passing through registers (LDS-DMA, 16 bytes per lane: waves 0 and 1
// fetch 128 nodes each); call once the ranking scratch is dead, with threads 0..127 present
__device__ __forceinline__ void tree_dense_fetch(const prism_replay_desc &rp, char *lds) {
    const int tid = threadIdx.x;
    if (tid < 128) {
        const float2 *src = tree_nodes(rp) + 256 + 2 * tid;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + TW_DENSE.off + (tid >> 6) * 1024), 16, 0, 0);
    }
}
// `walk_levels`: levels the walk covered; its last records lie in s_rec[(walk_levels & 1)].  `lds` as for the writer
// (the dense level is parked where the ranking keys were).
__device__ __forceinline__ void tree_dense_finish(const prism_replay_desc &rp, int32_t leaf, int rank, int cnt, int walk_levels,
                                                  char *lds) {
    const int tid = threadIdx.x;
    float2 *s_dense = reinterpret_cast<float2 *>(lds + TW_DENSE.off);   // [256] old values, parked by the caller
    const int4 *s_rec = reinterpret_cast<const int4 *>(lds + TW_REC.off) + (walk_levels & 1) * UPD_MAX;
    // the dense level's old values are in LDS (each fetching wave waits for its own LDS-DMA, then all meet) BEFORE any
    // touched node is written over its old value
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_only_barrier();
    if (tid < cnt) {             // (every thread of a run holds the same pair: several of them may store it)
        const int4 q = s_rec[rank];
        s_dense[(leaf >> walk_levels) - 256] = make_float2(__int_as_float(q.x), __int_as_float(q.y));
    }
    lds_only_barrier();
    if (tid < 64) {
        auto comb = [](float2 l, float2 r) { return make_float2(l.x + r.x, l.y < r.y ? l.y : r.y); };
        const float4 c01 = reinterpret_cast<const float4 *>(s_dense)[2 * tid];
        const float4 c23 = reinterpret_cast<const float4 *>(s_dense)[2 * tid + 1];
        const float2 p0 = comb(make_float2(c01.x, c01.y), make_float2(c01.z, c01.w));
        const float2 p1 = comb(make_float2(c23.x, c23.y), make_float2(c23.z, c23.w));
        reinterpret_cast<float4 *>(rp.tree)[64 + tid] = make_float4(p0.x, p0.y, p1.x, p1.y);      // nodes 128 + 2 tid, + 1
        float2 v = comb(p0, p1);
        tree_nodes(rp)[64 + tid] = v;
        // lane to lane: after step k the lanes with their k + 1 low bits clear hold node (64 >> (k + 1)) + (lane >> (k + 1))
#pragma unroll 1
        for (int k = 0; k < 6; ++k) {
            const float2 o = make_float2(__shfl_xor(v.x, 1 << k, 64), __shfl_xor(v.y, 1 << k, 64));
            v = ((tid >> k) & 1) ? comb(o, v) : comb(v, o);
            if ((tid & ((2 << k) - 1)) == 0) tree_nodes(rp)[(64 >> (k + 1)) + (tid >> (k + 1))] = v;
        }
    }
}

// all three in one workgroup: among equal leaves the highest t wins.  DENSE: the walk stops at the 256-node level and
// tree_dense_finish recomputes the rest (the caller has checked: levels >= TREE_DENSE_LEVELS, one pass, >= 128 threads);
// a template so that the shorter walk also carries fewer sibling registers (the hosts of this routine run 1024 threads
// at exactly 128 registers).
template <bool DENSE>
__device__ __forceinline__ void block_tree_write_impl(const prism_replay_desc &rp, int32_t my_idx, float my_val, int cnt,
                                                      char *lds, const float2 *__restrict__ sib, int sib_stride, int bd,
                                                      bool retire_idle) {
    constexpr int NL = DENSE ? TREE_MAX_LEVELS - TREE_DENSE_LEVELS : TREE_MAX_LEVELS;
    const bool active = (int)threadIdx.x < cnt;
    const int64_t cap = rp.tree_capacity;
    const int all_levels = 63 - __clzll((unsigned long long)cap);
    const int levels = DENSE ? all_levels - TREE_DENSE_LEVELS : all_levels;        // what the walk covers
    const int32_t leaf = active ? (int32_t)((int64_t)my_idx | cap) : (int32_t)cap;
    TREE_STAMP(4);
    SibRegs<NL> sr;
    tree_sib_prefetch<NL>(rp, leaf, active, levels, sib, sib_stride, sr);
    TREE_STAMP(23);
    const TreePrep p = tree_write_prepare(my_idx, leaf, my_val, cnt, all_levels, lds, bd);
    // (requested here: the ranking scratch they go to is dead, waves 0 and 1 are still present, and the walk's nine
    // levels hide the trip; they land before the walk stores that level itself -- every LDS read behind an LDS-DMA waits)
    if (DENSE) tree_dense_fetch(rp, lds);
    // the ranking is done (it splits its counting over up to four thread groups): from here on only the threads that
    // carry a leaf work, and every wave still present is one more wave at each of the `levels` barriers below
    if (retire_idle && (int)threadIdx.x >= ((cnt + 63) & ~63)) {
        if (DENSE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a leaving wave's own LDS-DMA has landed first)
        return;
    }
    tree_write_levels<NL>(rp, leaf, p, cnt, levels, lds, sr);
    if (DENSE) tree_dense_finish(rp, leaf, p.rank, cnt, levels, lds);
}
// (Both forms in ONE kernel push the 1024-thread hosts over their 128 registers: the host picks the kernel instantiation,
// tree_dense_ok below.)
__device__ __forceinline__ void block_tree_write(const prism_replay_desc &rp, int32_t my_idx, float my_val, int cnt, char *lds,
                                 const float2 *__restrict__ sib = nullptr, int sib_stride = 0, int bd = 0,
                                 bool retire_idle = false) {
    if (!bd) bd = (int)blockDim.x;
    block_tree_write_impl<false>(rp, my_idx, my_val, cnt, lds, sib, sib_stride, bd, retire_idle);
}
// host: may a single-workgroup update of `n` leaves by `threads` threads use the dense form?
inline bool tree_dense_ok(int64_t tree_capacity, int n, int threads) {
    int levels = 0;
    while (((int64_t)1 << (levels + 1)) <= tree_capacity) ++levels;
    return levels >= TREE_DENSE_LEVELS && threads >= 128 && n <= (threads < UPD_MAX ? threads : UPD_MAX);
}

// whole PrioritizedSampler.update_priority for one batch, executed by ONE workgroup (any size)
// `lds`: PER_UPDATE_LDS_BYTES of 16-byte aligned LDS supplied by the calling kernel (so that kernels
// hosting this routine as one role among others can alias it with their own scratch).
// `plan_out` (optional, needs n <= min(UPD_MAX, blockDim.x)): stop after the preparation and leave
// {value, lo | hi << 16, leaf, rank} per element there; per_update_finish completes the job (in a
// later kernel, so that the two halves hide behind different neighbours).
constexpr int PER_UPDATE_LDS_BYTES = TREE_WRITE_LDS_BYTES + 128;
constexpr LdsRegion PU_RED{TREE_WRITE_LDS_BYTES, 128, LDS_ALWAYS};          // float [<= 32] per-wave maxima, live across all passes
constexpr LdsRegion PU_REGIONS[] = {TW_SORTED, TW_VAL, TW_PART, TW_KEYS, TW_REC, TW_DENSE, PU_RED};
static_assert(lds_layout_ok(PU_REGIONS, PER_UPDATE_LDS_BYTES), "priority update: LDS regions overlap");
template <bool PREPARE_ONLY = false, bool DENSE = false>
__device__ __forceinline__ void per_update_block(const prism_replay_desc &rp, const int64_t *__restrict__ index,
                                 const float *__restrict__ priority, int n, float alpha, float eps, int take_abs,
                                 char *lds, const float2 *__restrict__ sib = nullptr, int sib_stride = 0,
                                 int4 *__restrict__ plan_out = nullptr, int live_threads = 0,
                                 const float *__restrict__ priority2 = nullptr) {
    // `priority2` (optional): the priority of element i is 0.5 priority[i] + 0.5 priority2[i] -- the TD error of a model with
    // both a distributional and a Q part, td = dl / 2 + ql / 2 (composite_model.py:135-137), taken from the two per-sample
    // losses where nobody has combined them yet (same expression, same bits)
    auto prio = [&](int i) __attribute__((always_inline)) {
        return priority2 ? priority[i] * 0.5f + priority2[i] * 0.5f : priority[i];
    };
    float *s_red = reinterpret_cast<float *>(lds + PU_RED.off);
    const int tid = threadIdx.x;
    // `live_threads` (a multiple of 64, >= n rounded up): the caller has already retired the waves above it -- a
    // workgroup that hosts this role with more waves than the batch needs pays for every one of them at each of the
    // ~25 workgroup barriers below (hardware barriers only count waves that have not ended)
    const int bd = live_threads ? live_threads : (int)blockDim.x;
    const int pass = min(UPD_MAX, bd);
    TREE_STAMP(0);
    // first pass's operands and the running max get in flight together
    const float old_max = tid == 0 ? rp.per_state[0] : 0.f;
    int32_t me = 0;
    float p0 = 0.f;
    if (tid < min(pass, n)) {
        p0 = prio(tid);
        if (take_abs) p0 = fabsf(p0);
        me = (int32_t)index[tid];
    }
    // running max of the raw priorities (torchrl tracks it before the +eps, **alpha)
    float m = tid < min(pass, n) ? p0 : -FLT_MAX;
    for (int i = pass + tid; i < n; i += bd) {
        float p = prio(i);
        if (take_abs) p = fabsf(p);
        m = fmaxf(m, p);
    }
    m = wave_max(m);
    if ((tid & 63) == 0) s_red[tid >> 6] = m;
    TREE_STAMP(1);
    for (int base = 0; base < n; base += pass) {
        const int cnt = min(pass, n - base);
        float val = 0.f;
        if (base) {
            __syncthreads();                                    // previous pass: stores drained, LDS free
            if (tid < cnt) {
                p0 = prio(base + tid);
                if (take_abs) p0 = fabsf(p0);
                me = (int32_t)index[base + tid];
            }
        }
        if (tid < cnt) val = pow_alpha(p0 + eps, alpha);
        TREE_STAMP(3);
        if (PREPARE_ONLY) {                                     // (the caller guarantees n <= pass)
            const int64_t cap = rp.tree_capacity;
            const int levels = 63 - __clzll((unsigned long long)cap);
            const int32_t leaf = tid < cnt ? (int32_t)((int64_t)me | cap) : (int32_t)cap;
            const TreePrep p = tree_write_prepare(me, leaf, val, cnt, levels, lds, bd);
            if (tid < cnt) plan_out[tid] = make_int4(__float_as_int(p.val), p.lo | (p.hi << 16), leaf, p.rank);
        } else {
            // (a sibling record is only valid for one pass; DENSE: the host has checked that there is only one)
            block_tree_write_impl<DENSE>(rp, me, val, cnt, lds, n <= pass ? sib : nullptr, sib_stride, bd,
                                         live_threads != 0 && n <= pass);
        }
        if (base == 0 && tid == 0) {
            float mm = old_max;                                 // (s_red was published before the first barrier)
            for (int w = 0; w < (bd >> 6); ++w) mm = fmaxf(mm, s_red[w]);
            rp.per_state[0] = mm;
        }
    }
}

// second half of a prepared update: `plan` as left by per_update_block(plan_out); n <= blockDim.x.
// `sib_state` (optional): 1 = `sib` holds the siblings of exactly these paths; consumed (reset to 0).
__device__ __forceinline__ void per_update_finish(const prism_replay_desc &rp, const int4 *__restrict__ plan, int n, char *lds,
                                  const float2 *__restrict__ sib, int sib_stride, unsigned int *sib_state) {
    const int tid = threadIdx.x;
    const bool active = tid < n;
    const int64_t cap = rp.tree_capacity;
    const int levels = 63 - __clzll((unsigned long long)cap);
    // everything this block needs from global memory is requested here, in one go: the plan, the
    // record state and (speculatively) the recorded siblings
    const int4 pl = active ? plan[tid] : make_int4(0, 0, (int)cap, 0);
    const unsigned int rec = sib_state ? *sib_state : 0u;
    SibRegs<TREE_MAX_LEVELS> sr;
    if (sib) tree_sib_prefetch(rp, pl.z, active, levels, sib, sib_stride, sr);
    if (!sib || rec != 1u) tree_sib_prefetch(rp, pl.z, active, levels, nullptr, 0, sr);       // uniform; rare
    TreePrep p;
    p.val = __int_as_float(pl.x);
    p.lo = pl.y & 0xffff;
    p.hi = pl.y >> 16;
    p.rank = pl.w;
    tree_write_levels(rp, pl.z, p, n, levels, lds, sr);
    if (sib_state && tid == 0) *sib_state = 0u;
}

template <bool DENSE>
static __global__ __launch_bounds__(1024) void per_update_kernel(prism_replay_desc rp, const int64_t *__restrict__ index,
                                                         const float *__restrict__ priority, int n,
                                                         float alpha, float eps, int take_abs) {
    __shared__ __attribute__((aligned(16))) char s_pool[PER_UPDATE_LDS_BYTES];
    per_update_block<false, DENSE>(rp, index, priority, n, alpha, eps, take_abs, s_pool);
}

// rows of an insert batch -> ring slots (any number of workgroups)
static __global__ void replay_store_rows_kernel(prism_replay_desc rp, int n, const int32_t *__restrict__ slots,
                                         const float *__restrict__ obs, const float *__restrict__ succ_obs,
                                         const float *__restrict__ reward, const int32_t *__restrict__ action,
                                         const uint8_t *__restrict__ flags) {
    const int O = rp.obs_elems;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t s = slots[i];
        float *d0 = rp.obs + s * O, *d1 = rp.succ_obs + s * O;
        const float *s0 = obs + (int64_t)i * O, *s1 = succ_obs + (int64_t)i * O;
        if ((O & 3) == 0) {
            for (int k = threadIdx.x; k < O / 4; k += blockDim.x) {
                reinterpret_cast<float4 *>(d0)[k] = reinterpret_cast<const float4 *>(s0)[k];
                reinterpret_cast<float4 *>(d1)[k] = reinterpret_cast<const float4 *>(s1)[k];
            }
        } else {
            for (int k = threadIdx.x; k < O; k += blockDim.x) {
                d0[k] = s0[k];
                d1[k] = s1[k];
            }
        }
        if (threadIdx.x == 0) {
            rp.reward[s] = reward[i];
            rp.action[s] = action[i];
            rp.flags[s] = flags[i];
        }
    }
}

// links + default priority.  The reference applies the rows one after the other; for a batch of
// consecutive ring slots (what a writer cursor produces) whose predecessors are not overwritten later
// in the same batch, three parallel phases give the same final state:
//   A  detach the old neighbours of every overwritten slot (reads the state before the batch),
//   B  reset the slots' own link/back,   C  attach each row to its predecessor.
// Anything else (arbitrary slot lists, a batch that wraps over its own predecessors) takes the
// sequential path.
static __global__ __launch_bounds__(1024) void replay_link_kernel(prism_replay_desc rp, int n,
                                                          const int32_t *__restrict__ slots,
                                                          const int32_t *__restrict__ prev_slot, float alpha,
                                                          float eps) {
    __shared__ __attribute__((aligned(16))) char s_pool[TREE_WRITE_LDS_BYTES];
    __shared__ int s_serial;
    const int tid = threadIdx.x, bd = blockDim.x;
    if (tid == 0) s_serial = 0;
    __syncthreads();
    {
        const int64_t first = slots[0], cap = rp.capacity;
        bool bad = false;
        for (int i = tid; i < n; i += bd) {
            if (slots[i] != (int32_t)((first + i) % cap)) bad = true;
            const int32_t p = prev_slot[i];
            if (p >= 0) {
                const int64_t pos = ((int64_t)p - first + cap) % cap;
                if (pos < n && pos >= i) bad = true;        // predecessor overwritten at or after this row
            }
        }
        if (bad) s_serial = 1;
    }
    __syncthreads();
    if (s_serial) {
        if (tid == 0) {
            for (int i = 0; i < n; ++i) {
                const int32_t s = slots[i];
                const int32_t b = rp.back[s];
                if (b >= 0 && rp.link[b] == s) rp.link[b] = -1;   // predecessor of the overwritten row
                const int32_t q = rp.link[s];
                if (q >= 0 && rp.back[q] == s) rp.back[q] = -1;   // successor of the overwritten row
                rp.link[s] = -1;
                rp.back[s] = -1;
                const int32_t p = prev_slot[i];
                if (p >= 0) {
                    rp.link[p] = s;
                    rp.back[s] = p;
                }
            }
        }
    } else {
        for (int i = tid; i < n; i += bd) {                      // A
            const int32_t s = slots[i];
            const int32_t b = rp.back[s], q = rp.link[s];
            if (b >= 0 && rp.link[b] == s) rp.link[b] = -1;
            if (q >= 0 && rp.back[q] == s) rp.back[q] = -1;
        }
        __syncthreads();
        for (int i = tid; i < n; i += bd) {                      // B
            const int32_t s = slots[i];
            rp.link[s] = -1;
            rp.back[s] = -1;
        }
        __syncthreads();
        for (int i = tid; i < n; i += bd) {                      // C
            const int32_t p = prev_slot[i];
            if (p >= 0) {
                rp.link[p] = slots[i];
                rp.back[slots[i]] = p;
            }
        }
    }
    if (!rp.tree) return;
    const float prio = pow_alpha(rp.per_state[0] + eps, alpha);
    const int pass = min(UPD_MAX, (int)blockDim.x);
    for (int base = 0; base < n; base += pass) {
        const int cnt = min(pass, n - base);
        __syncthreads();
        const int32_t me = (int)threadIdx.x < cnt ? (int32_t)slots[base + threadIdx.x] : 0;
        block_tree_write(rp, me, prio, cnt, s_pool);
    }
}

// internal nodes of one level from their children (used after leaves were written in bulk)
static __global__ void per_rebuild_level_kernel(prism_replay_desc rp, int64_t first, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int64_t p = first + i;
        const float4 ch = reinterpret_cast<const float4 *>(rp.tree)[p];     // {left.sum, left.min, right.sum, right.min}
        tree_nodes(rp)[p] = make_float2(ch.x + ch.z, ch.y < ch.w ? ch.y : ch.w);
    }
}

// SumSegmentTree::ScanLowerBound with fp32 subtract-as-you-go; nodes below `top` come from LDS
__device__ __forceinline__ int64_t tree_descend(const prism_replay_desc &rp, const float *s_top, int64_t top,
                                                float mass) {
    const int64_t cap = rp.tree_capacity;
    if (mass > s_top[1]) return rp.capacity;
    int64_t node = 1;
    float v = mass;
    while (node < cap) {
        node <<= 1;
        const float lv = node < top ? s_top[node] : tree_nodes(rp)[node].x;
        if (v > lv) {
            v -= lv;
            node |= 1;
        }
    }
    return node ^ cap;
}

// The same descent over the interleaved tree with a float2 LDS top, which also records for every
// level s (0 = leaf level) {sum, min} of the sibling of the node it steps into: sib[s * stride]
// (give it LDS: a global store issued between the loads of the lower levels makes every later load
// wait for the store's acknowledgement as well -- loads and stores share one in-order counter).
__device__ __forceinline__ int64_t tree_descend_record(const prism_replay_desc &rp, const float2 *s_top, int64_t top,
                                                       float mass, float2 *sib, int stride, float *leaf_sum) {
    const int64_t cap = rp.tree_capacity;
    if (mass > s_top[1].x) return rp.capacity;
    int64_t node = 1;
    float v = mass;
    int s = 63 - __clzll((unsigned long long)cap);
    auto step = [&](const float4 ch) {      // ch = {left.sum, left.min, right.sum, right.min} of `node`'s children
        node <<= 1;
        --s;
        const bool right = v > ch.x;
        if (right) {
            v -= ch.x;
            node |= 1;
        }
        sib[(int64_t)s * stride] = right ? make_float2(ch.x, ch.y) : make_float2(ch.z, ch.w);
        *leaf_sum = right ? ch.z : ch.x;
    };
    while (node < cap && 2 * node < top) step(*reinterpret_cast<const float4 *>(&s_top[2 * node]));
    // Below the LDS top every level costs a full memory round trip, so fetch three levels per trip:
    // the 2 + 4 + 8 descendants of `node` are three contiguous, aligned runs (16 + 32 + 64 bytes,
    // the last exactly one cache line) whose addresses are known before any of them arrives.
    const float4 *t4 = reinterpret_cast<const float4 *>(rp.tree);
    while (node < cap) {
        const int64_t n = node;
        const bool two = 2 * n < cap, three = 4 * n < cap;
        const float4 a0 = t4[n];
        float4 b0, b1, c0, c1, c2, c3;
        if (two) {
            b0 = t4[2 * n];
            b1 = t4[2 * n + 1];
        }
        if (three) {
            c0 = t4[4 * n];
            c1 = t4[4 * n + 1];
            c2 = t4[4 * n + 2];
            c3 = t4[4 * n + 3];
        }
        // (selections spelled as two-way register selects: an indexed pick lands in scratch memory)
        auto sel = [](bool hi, const float4 &x, const float4 &y) {
            return make_float4(hi ? y.x : x.x, hi ? y.y : x.y, hi ? y.z : x.z, hi ? y.w : x.w);
        };
        step(a0);
        const bool r1 = node & 1;
        if (two) step(sel(r1, b0, b1));
        const bool r2 = node & 1;
        if (three) step(sel(r1, sel(r2, c0, c1), sel(r2, c2, c3)));
    }
    return node ^ cap;
}

// The descent as ONE WAVE with one sample (the fused front launch): every lane runs it with the same mass.  The node
// pairs {left.sum, left.min, right.sum, right.min} = t4[parent] live in registers, spread over the lanes, and the walk
// picks them up with v_readlane at a wave-uniform lane number -- no LDS, no per-level memory round trip.
// A dependent round trip to the tree costs ~1.8 k cycles here (the L2s are invalidated at every kernel boundary), a level
// out of registers ~100.  `top[j]` = t4[lane + 64 j], j < 2: the SEVEN levels under the root, fetched by the caller together
// with its other first-round loads; below them a trip fetches the up to SIX levels under the current node, one load per
// lane: pair number i = lane (1 <= i < 64) is the node's descendant i - 2^k at depth k = floor(log2 i).  17 levels: two
// dependent trips (the one-lane form made four of three levels each).  Measured and not kept: nine levels up front and
// eight per trip (one dependent trip for 17 levels) -- the 64 cold lines of the nine-level top cost the first round of
// loads 2.4 k cycles more than the trip saved.
// Same arithmetic, same order as tree_descend_record.  The sibling {sum, min} of level s (0 = leaf level) is left in
// lane s of `sib`.
struct WaveDescent {
    int64_t idx;
    float leaf_sum;
    float2 sib;
};
__device__ __forceinline__ float4 readlane4(const float4 &v, int lane) {
    return make_float4(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), lane)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), lane)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), lane)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.w), lane)));
}
__device__ __forceinline__ float4 pick4(bool hi, const float4 &x, const float4 &y) {
    return make_float4(hi ? y.x : x.x, hi ? y.y : x.y, hi ? y.z : x.z, hi ? y.w : x.w);
}
constexpr int WAVE_TOP_REGS = 2, WAVE_TOP_LEVELS = 7, WAVE_TRIP_LEVELS = 6;
__device__ __forceinline__ void wave_top_fetch(const prism_replay_desc &rp, float4 (&top)[WAVE_TOP_REGS]) {
    const float4 *t4 = reinterpret_cast<const float4 *>(rp.tree);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < WAVE_TOP_REGS; ++j) {
        top[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane + 64 * j < rp.tree_capacity) top[j] = t4[lane + 64 * j];
    }
}
__device__ __forceinline__ WaveDescent tree_descend_wave(const prism_replay_desc &rp, const float4 (&top)[WAVE_TOP_REGS],
                                                         float mass) {
    const int lane = threadIdx.x & 63;
    const int cap = (int)rp.tree_capacity;            // <= 2^24
    WaveDescent r;
    r.sib = make_float2(0.f, 0.f);
    r.leaf_sum = 0.f;
    const float root = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(top[0].z), 0));      // node 1 = second half of t4[0]
    if (mass > root) {
        r.idx = rp.capacity;
        return r;
    }
    const int L = 31 - __clz(cap);
    // everything but `v` is kept scalar: the comparison's lane mask (all lanes agree) selects between readlane results in
    // SGPRs
    int node = 1, s = L, sibx = 0, siby = 0;
    float v = mass, leaf = 0.f;
    auto step = [&](const float4 ch) {
        --s;
        const bool right = __ballot(v > ch.x) != 0ull;
        v = right ? v - ch.x : v;
        node = 2 * node + (right ? 1 : 0);
        const float sx = right ? ch.x : ch.z, sy = right ? ch.y : ch.w;
        sibx = lane == s ? __float_as_int(sx) : sibx;
        siby = lane == s ? __float_as_int(sy) : siby;
        leaf = right ? ch.z : ch.x;
    };
    if (L >= WAVE_TOP_LEVELS) {
#pragma unroll
        for (int k = 0; k < 6; ++k) step(readlane4(top[0], node));       // node in [2^k, 2^(k+1))
        step(readlane4(top[1], node - 64));
    } else {
        for (int k = 0; k < L; ++k) step(readlane4(node < 64 ? top[0] : top[1], node & 63));
    }
    const float4 *t4 = reinterpret_cast<const float4 *>(rp.tree);
    const int kd = 31 - __clz(lane | 1);
    while (s > 0) {
        const int nl = s < WAVE_TRIP_LEVELS ? s : WAVE_TRIP_LEVELS;       // s = levels still to go
        float4 sub = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane >= 1 && lane < (1 << nl)) sub = t4[(node << kd) + (lane - (1 << kd))];
        const int base = node;
        for (int j = 1; j <= nl; ++j) step(readlane4(sub, node - (base << (j - 1)) + (1 << (j - 1))));
    }
    r.idx = node ^ cap;
    r.sib = make_float2(__int_as_float(sibx), __int_as_float(siby));
    r.leaf_sum = leaf;
    return r;
}

// _compute_n_step (timestep_buffer.py:198-238) over the slot arrays; wave-uniform
struct NStepResult {
    int64_t last;      // slot where the walk stopped
    double ret;        // sum_k reward_k * gamma**k (float64 accumulate)
    double gamma;      // gamma ** (number of rewards summed)
    uint32_t flags;    // flags of the last slot
};
__device__ __forceinline__ NStepResult nstep_walk(const prism_replay_desc &rp, int64_t first) {
    NStepResult r;
    int64_t cur = first;
    double ret = 0.0, gamma = 1.0;
    // reward, flags AND link of a slot are requested together: one memory round trip per hop (asking for the link only
    // once the flags say it is wanted made it two)
    uint32_t f = 0;
    for (int k = 0; k < rp.n_step; ++k) {
        const float rw = rp.reward[cur];
        const int32_t nx = rp.link[cur];
        f = rp.flags[cur];
        ret += (double)rw * rp.gammas[k];
        gamma = rp.gammas[k + 1];
        const bool incomplete = (k != rp.n_step - 1);
        if ((f & PRISM_FLAG_HAS_NEXT) && !(f & PRISM_FLAG_TRUNC) && incomplete && nx >= 0)
            cur = nx;
        else
            break;
    }
    r.last = cur;
    r.ret = ret;
    r.gamma = gamma;
    r.flags = f;            // (flags of the slot the walk stopped at: the last ones read)
    return r;
}

// ---- PER sample: one lane per sample, tree top in LDS --------------------------------------
static __global__ __launch_bounds__(256) void per_sample_kernel(prism_replay_desc rp, int64_t size, int batch,
                                                        const float *__restrict__ mass_in, uint64_t seed,
                                                        uint64_t offset, float beta,
                                                        int64_t *__restrict__ out_index,
                                                        float *__restrict__ out_weight) {
    __shared__ float s_top[TOP_NODES];
    __shared__ float s_scratch[128];
    const int64_t cap = rp.tree_capacity;
    const int64_t top = cap < TOP_NODES ? cap : TOP_NODES;   // nodes [1, top) are internal or leaves of a tiny tree
#pragma unroll 8
    for (int i = threadIdx.x; i < top; i += blockDim.x) s_top[i] = tree_nodes(rp)[i].x;
    const float p_sum = block_tree_query<false>(tree_nodes(rp), cap, rp.capacity, size, s_scratch);
    const float p_min = block_tree_query<true>(tree_nodes(rp), cap, rp.capacity, size, s_scratch);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        rp.per_state[1] = p_sum;
        rp.per_state[2] = p_min;
        int st = 0;
        if (!(p_sum > 0.0f)) st |= PRISM_STATUS_NONPOSITIVE_PSUM;
        if (!(p_min > 0.0f)) st |= PRISM_STATUS_NONPOSITIVE_PMIN;
        if (st) atomicOr(rp.status, st);
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    float mass;
    if (mass_in) {
        mass = mass_in[i];
    } else {
        uint32_t r[4];
        Philox ph(seed);
        ph(offset + (uint64_t)i, 0x5045524dull /* "PERM" */, r);
        mass = (float)(0.0 + ((double)p_sum - 0.0) * u64_to_unit_double(r[0], r[1]));
    }
    int64_t idx = tree_descend(rp, s_top, top, mass);
    if (idx > size - 1) idx = size - 1;
    out_index[i] = idx;
    const float leaf = tree_nodes(rp)[idx | cap].x;
    out_weight[i] = pow_neg_beta(leaf / p_min, beta);
}

static __global__ void per_query_kernel(prism_replay_desc rp, int64_t size, float *out2) {
    __shared__ float s_scratch[128];
    const float p_sum = block_tree_query<false>(tree_nodes(rp), rp.tree_capacity, rp.capacity, size, s_scratch);
    const float p_min = block_tree_query<true>(tree_nodes(rp), rp.tree_capacity, rp.capacity, size, s_scratch);
    if (threadIdx.x == 0) {
        out2[0] = p_sum;
        out2[1] = p_min;
    }
}

static __global__ void uniform_sample_kernel(int64_t size, int batch, uint64_t seed, uint64_t offset,
                                      int64_t *__restrict__ out_index) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    uint32_t r[4];
    Philox ph(seed);
    ph(offset + (uint64_t)i, 0x554e4946ull /* "UNIF" */, r);
    const uint64_t x = ((uint64_t)r[0] << 32) | r[1];
    out_index[i] = (int64_t)__umul64hi(x, (uint64_t)size);
}

// ---- n-step walk + row gather: one workgroup (128 lanes) per sampled slot ----------------------
static __global__ __launch_bounds__(128) void replay_gather_kernel(prism_replay_desc rp, const int64_t *__restrict__ index,
                                                           int batch, float *__restrict__ out_obs,
                                                           float *__restrict__ out_next_obs,
                                                           float *__restrict__ out_reward,
                                                           uint8_t *__restrict__ out_nonterminal,
                                                           float *__restrict__ out_gamma,
                                                           int64_t *__restrict__ out_action) {
    const int b = blockIdx.x;
    if (b >= batch) return;
    const int64_t first = index[b];
    const NStepResult ns = nstep_walk(rp, first);   // wave-uniform: every lane runs it
    const int64_t cur = ns.last;
    const uint32_t fl = ns.flags;
    const double ret = ns.ret, gamma = ns.gamma;
    const int O = rp.obs_elems;
    const float *src_obs = rp.obs + first * O;
    const float *src_next = (fl & PRISM_FLAG_HAS_NEXT) ? rp.succ_obs + cur * O : src_obs;
    float *d0 = out_obs + (int64_t)b * O, *d1 = out_next_obs + (int64_t)b * O;
    if ((O & 3) == 0) {
        for (int k = threadIdx.x; k < O / 4; k += blockDim.x) {
            const float4 a = reinterpret_cast<const float4 *>(src_obs)[k];
            const float4 c = reinterpret_cast<const float4 *>(src_next)[k];
            reinterpret_cast<float4 *>(d0)[k] = a;
            reinterpret_cast<float4 *>(d1)[k] = c;
        }
    } else {
        for (int k = threadIdx.x; k < O; k += blockDim.x) {
            d0[k] = src_obs[k];
            d1[k] = src_next[k];
        }
    }
    if (threadIdx.x == 0) {
        out_reward[b] = (float)ret;
        out_nonterminal[b] = (fl & PRISM_FLAG_DONE) ? 0 : 1;
        out_gamma[b] = (float)gamma;
        out_action[b] = (int64_t)rp.action[first];
    }
}

}  // namespace prism
