// Fused kernels of the whole learner step (prism/learner.py:95-125), cutting the launch count from
// ten to six: front (PER sample + n-step gather + conv embed), tile_fwd, loss, bwd,
// post (slab reduce + small tensors + conv backward), back (clip + Adam + priority writeback).
// Every kernel here is a thin role dispatcher over the block-level routines of iqn_kernels.h /
// replay_kernels.h, so fused and unfused paths execute the same arithmetic.
#pragma once
#include "iqn_kernels.h"
#include "qhead_kernels.h"
#include "replay_kernels.h"

namespace prism {


// ---- the parameter-only roles that ride along with the embed / front launch ---------------------
// IQN: per weight set (online, target) u,v (H/4 blocks, LayerNorm only) and the stream-packed weights;
// Q heads: per head and weight set W1 packing and u_h,v_h; per head ||theta_h||^2 (online).
struct ExtraDims {
    int use_iqn, n_heads, has_target, head_layers, Hi, Hq, ln, split, cos_tiles;
};
__host__ __device__ inline ExtraDims extra_dims(const IqnArgs &a) {
    return ExtraDims{a.use_iqn, a.n_heads, a.has_target, a.head_layers, a.Hi, a.Hq, a.ln, a.split, a.cos_tiles};
}
__host__ __device__ inline int iqn_pack_blocks_for(int H, int split) { return split ? iqn_pack_split_blocks(H) : iqn_pack_blocks(H); }
__host__ __device__ inline int q_pack_blocks_for(int H, int split) { return split ? q_pack_split_blocks_per_head(H) : q_pack_blocks_per_head(H); }
__host__ __device__ inline int front_extra_blocks(const ExtraDims &d) {
    const int sets = 1 + (d.has_target ? 1 : 0);
    const int heads = d.head_layers == 1 ? 0 : d.n_heads;        // single-Linear DQN head: nothing to pack or precompute
    int n = 0;
    if (d.use_iqn) n += sets * ((d.ln ? d.Hi / 4 : 0) + iqn_pack_blocks_for(d.Hi, d.split));
    n += heads * (sets * (q_pack_blocks_for(d.Hq, d.split) + (d.ln ? d.Hq / 4 : 0)) + Q_NORM_PARTS);
    n += d.cos_tiles;          // (last: one block per IQN forward tile, cos_basis_block)
    return n;
}

// Quantile samples and cos basis of ONE forward tile (iqn_model.py:90-92), prepared beside the sampling / convolution
// workgroups of the embed / front launch instead of in the forward tile's prologue -- there the draw (a cold counter word,
// ten Philox rounds), two cosines a thread, the split into bf16 pieces and two workgroup barriers sat in front of the
// first matrix instruction of every tile (7 k cycles of a 45 k tile, stamped).  Same expressions, same bits as the tile's own
// prologue (fwd_kernels.h), which stays for the passes that do not come through here (acting).
// 256 threads: row m = tid >> 4, basis pairs (tid & 15) and (tid & 15) + 16.
__device__ __forceinline__ void cos_basis_block(const IqnArgs &a, int x) {
    const int tid = threadIdx.x;
    int tile = x, pi = -1;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const bool mine = i < a.n_pass && a.pass[i].kind != 1 && a.pass[i].cospk != nullptr;
        const int nt = mine ? a.pass[i].n_tiles : 0;
        const bool hit = mine && pi < 0 && tile < nt;
        pi = hit ? i : pi;
        tile -= (pi < 0) ? nt : 0;
    }
    if (pi < 0) return;
    // (fields picked with comparisons, not a.pass[pi]: a run-time index into the by-value argument struct moves it to scratch)
    const float *tau_in = nullptr, *tau_in2 = nullptr;
    const unsigned int *cpk = nullptr;
    int T = 1, save = 0, sid0 = 0, kind = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i == pi) {
            tau_in = a.pass[i].tau_in; tau_in2 = a.pass[i].tau_in2; cpk = a.pass[i].cospk;
            T = a.pass[i].T; save = a.pass[i].save; sid0 = a.pass[i].stream_id; kind = a.pass[i].kind;
        }
    const int m = tid >> 4, kq = tid & 15, tsh = 31 - __clz(T);
    int b, t, nx = 0;
    int64_t srow;
    if (kind == 0) {
        const int row = tile * 16 + m;
        b = min((T & (T - 1)) == 0 ? row >> tsh : row / T, a.B - 1);
        t = row - b * T;
        srow = save ? (int64_t)row : -1;
    } else {
        const int s = m >> (tsh + 1), j = m & (2 * T - 1);
        b = tile * (16 >> (tsh + 1)) + s;
        nx = j >= T;
        t = j & (T - 1);
        srow = nx ? -1 : (int64_t)b * T + t;
    }
    const float *tin = (kind == 2 && nx) ? tau_in2 : tau_in;
    const int sid = (kind == 2 && nx) ? 1 : sid0;
    float tau;
    if (tin) {
        tau = tin[min((int64_t)t, (int64_t)T - 1) * a.Bt + b];
    } else {
        uint32_t rr[4];
        Philox ph(a.seed);
        ph(a.offset + (a.rng ? a.rng[1] : 0ull) + (uint64_t)((int64_t)t * a.Bt + b), 0x54415530ull + (uint64_t)sid, rr);
        tau = u32_to_unit_float(rr[0]);
    }
    unsigned int *blk = const_cast<unsigned int *>(cpk) + (size_t)tile * CP_TILE;
    if (kq == 0) {
        if (a.tau_out && t < T) a.tau_out[(int64_t)sid * a.maxT * a.Bt + (int64_t)t * a.Bt + b] = tau;
        blk[3 * 16 * 32 + m] = __float_as_uint(tau);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int kp = kq + 16 * q, k0 = 2 * kp;
        const float c0 = cosf((tau * (float)(k0 + 1)) * PI_F), c1 = cosf((tau * (float)(k0 + 2)) * PI_F);
        if (srow >= 0) {
            __builtin_nontemporal_store(c0, &a.ws.cosb[srow * K_BASIS + k0]);
            __builtin_nontemporal_store(c1, &a.ws.cosb[srow * K_BASIS + k0 + 1]);
        }
        const unsigned int h = pack_bf16(c0, c1);
        const float ra = c0 - __uint_as_float(h << 16), rb = c1 - __uint_as_float(h & 0xffff0000u);
        const unsigned int md = pack_bf16(ra, rb);
        const float sa = ra - __uint_as_float(md << 16), sb = rb - __uint_as_float(md & 0xffff0000u);
        blk[(0 * 16 + m) * 32 + kp] = h;
        blk[(1 * 16 + m) * 32 + kp] = md;
        blk[(2 * 16 + m) * 32 + kp] = pack_bf16(sa, sb);
    }
}

// The TARGET network only changes when it is synchronised: its packed copies and u/v are rebuilt only while
// ws.ticket[2] ("target set packed") is zero -- the post launch of every update sets it, whoever writes target_params
// (prism_sync_target's caller, a checkpoint load) clears it.  With ten Q heads that is half of this launch's blocks.
__device__ __forceinline__ void front_extra_block(const IqnArgs &a, int x, float *s_red) {
    {
        ExtraDims d0 = extra_dims(a);
        d0.cos_tiles = 0;
        const int n_other = front_extra_blocks(d0);
        if (x >= n_other) {
            cos_basis_block(a, x - n_other);
            return;
        }
    }
    const int tid = threadIdx.x;
    const int sets = 1 + (a.has_target ? 1 : 0);
    const bool target_done = a.has_target && a.ws.ticket[2] != 0u;
    if (a.use_iqn) {
        const int nuv = a.ln ? a.Hi / 4 : 0, npk = iqn_pack_blocks_for(a.Hi, a.split);
        for (int set = 0; set < sets; ++set) {
            if (x < nuv) {
                if (set && target_done) return;
                iqn_uv_block(a, set, x * 4 + (tid >> 6), tid & 63);
                return;
            }
            x -= nuv;
            if (x < npk) {
                if (set && target_done) return;
                const float *Pp = set ? a.target_params : a.params;
                if (a.split)
                    pack_split_block(Pp + a.off.phi_w, Pp + a.off.iqn_w1, a.ln ? Pp + a.off.iqn_ln1_g : nullptr, a.Hi, true,
                                     a.ws.wpk[set], x, tid);
                else
                    pack_weights_block(Pp, a.off, a.Hi, a.ln, a.ws.wpk[set], x, tid);
                return;
            }
            x -= npk;
        }
    }
    const int nuv = a.ln ? a.Hq / 4 : 0, npk = q_pack_blocks_for(a.Hq, a.split);
    const int per_head = sets * (npk + nuv) + Q_NORM_PARTS;
    const int hd = x / per_head;
    x -= hd * per_head;
    for (int set = 0; set < sets; ++set) {
        if (x < npk) {
            if (set && target_done) return;
            const float *Pp = set ? a.target_params : a.params;
            if (a.split) {
                const float *Ph = Pp + a.off.head_base + (int64_t)hd * a.off.head_stride;
                pack_split_block(nullptr, Ph + a.off.h_w1, a.ln ? Ph + a.off.h_ln1_g : nullptr, a.Hq, false,
                                 a.ws.q_wpk[set] + (size_t)hd * q_pack_split_floats(a.Hq), x, tid);
            } else {
                pack_head_w1_block(Pp, a.off, a.Hq, a.ln, a.ws.q_wpk[set], hd, x, tid);
            }
            return;
        }
        x -= npk;
        if (x < nuv) {
            if (set && target_done) return;
            q_uv_block(a, set, hd, x * 4 + (tid >> 6), tid & 63);
            return;
        }
        x -= nuv;
    }
    q_head_norm_block(a, hd, x, s_red);
}

__device__ void embed_extra_block(const IqnArgs &a, int x, float *s_red) { front_extra_block(a, x, s_red); }

struct FrontArgs {
    int64_t size;          // stored items
    const float *mass;     // [B] or NULL -> Philox
    uint64_t seed, offset;
    const uint64_t *rng;   // device counters or NULL
    float beta;
    int use_per;
    int64_t *out_index;
    float *out_weight;
    // minibatch outputs (the static batch)
    float *obs, *next_obs, *reward, *gamma;
    uint8_t *nonterminal;
    int64_t *action;
};

// blocks [0, B): one sample each — tree descent, n-step walk, row gather, conv3x3+ReLU of both
// observations;  blocks [B, B + H/4): u = W1 g1, v = W1 beta1;  then PACK_BLOCKS (x2 with a target
// network) blocks refresh the fragment-packed weight copies tile_fwd streams.
// 256 threads: wave 0 samples, waves 1-3 stage the conv weights; the convolutions run on the matrix core, one 16-position
// tile per wave (conv_embed_rows).  (An eight-wave form that convolved with fmaf chains beside the n-step walk was the
// round-3 intermediate: the MFMA convolution is shorter on four waves than that one was on eight.)
__global__ __launch_bounds__(256) void step_front_kernel(IqnArgs a, prism_replay_desc rp, FrontArgs f) {
    kernarg_prefetch<sizeof(IqnArgs) + sizeof(prism_replay_desc) + sizeof(FrontArgs)>();
    __shared__ __attribute__((aligned(16))) float s_scratch[256];
    __shared__ __attribute__((aligned(16))) float s_obs[2][1000];
    __shared__ __attribute__((aligned(16))) float s_w[2][CONV_W_FLOATS];
    __shared__ float s_b[2][16];
    __shared__ int64_t s_i64[2];
    __shared__ float2 s_sibrec[TREE_MAX_LEVELS];
    __shared__ float s_out_w;
    __shared__ unsigned int s_rec_state, s_rec_state2;
    __shared__ uint32_t s_flags;
    const int B = a.B, C = a.C, tid = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= B) {
        PRISM_STAMP(27);
        front_extra_block(a, b - B, s_scratch);
        PRISM_STAMP(31);
        return;
    }
    PRISM_STAMP(27);
    // Everything that does not depend on the sampled index is requested in ONE round of loads.  Wave 0 is the sampler:
    // its lanes fetch the tree top and the nodes of the p_sum / p_min query and go on alone, everything in registers;
    // waves 1-3 stage the conv weights of both networks meanwhile -- those are not needed before the rows are convolved.
    const float *P0 = a.params, *P1 = a.has_target ? a.target_params : a.params;
    const int64_t cap = rp.tree_capacity;
    int64_t idx;
    float w_leaf = 1.f, w_pmin = 1.f;
    unsigned int w_rec = 1u;
    float2 w_sib = make_float2(0.f, 0.f);
    if (tid >= 64) {
        conv_w_to_lds(P0, a.off, C, s_w[0], s_b[0], tid - 64, 192);
        conv_w_to_lds(P1, a.off, C, s_w[1], s_b[1], tid - 64, 192);
    } else if (f.use_per) {
        // wave 0, every lane with the same values: node pairs of the seven levels under the root and the query nodes in
        // one round of loads, the Philox draw while they fly, then the whole descent out of registers (tree_descend_wave)
        float4 top[WAVE_TOP_REGS];
        float mass_in = 0.f;
        uint64_t ctr = 0;
        if (f.mass) mass_in = f.mass[b];
        else if (f.rng) ctr = f.rng[0];               // (the draw hangs on this word: asked for first)
        wave_top_fetch(rp, top);
        const float2 qv = tree_query_fetch_wave(tree_nodes(rp), cap, rp.capacity, f.size);
        double unit = 0.0;
        if (!f.mass) {
            uint32_t r[4];
            Philox ph(f.seed);
            ph(f.offset + ctr + (uint64_t)b, 0x5045524dull, r);
            unit = u64_to_unit_double(r[0], r[1]);
        }
        PRISM_STAMP(24);
        const float2 pq = tree_query_fold_wave(qv);
        const float p_sum = pq.x, p_min = pq.y;
        if (b == 0 && tid == 0) {
            rp.per_state[1] = p_sum;
            rp.per_state[2] = p_min;
            int st = 0;
            if (!(p_sum > 0.0f)) st |= PRISM_STATUS_NONPOSITIVE_PSUM;
            if (!(p_min > 0.0f)) st |= PRISM_STATUS_NONPOSITIVE_PMIN;
            if (st) atomicOr(rp.status, st);
        }
        PRISM_STAMP(28);
        PRISM_STAMP2(0);
        const float mass = f.mass ? mass_in : (float)(0.0 + ((double)p_sum - 0.0) * unit);
        // the descent reads both children of every path node; the one it does not step into is exactly what the
        // priority writeback needs later -> recorded, level s in lane s
        WaveDescent wd = tree_descend_wave(rp, top, mass);
        idx = wd.idx;
        PRISM_STAMP2(1);
        float leaf_sum = wd.leaf_sum;
        unsigned int rec = 1u;                                  // 1 = record valid, 2 = not usable
        if (idx > f.size - 1) {
            idx = f.size - 1;
            leaf_sum = tree_nodes(rp)[idx | cap].x;
            rec = 2u;
        }
        if (tid == 0) s_i64[0] = idx;
        // (importance weight and sibling record: nobody needs them before the rows are convolved -> behind the barrier,
        // beside the flight of the observation row)
        w_leaf = leaf_sum;
        w_pmin = p_min;
        w_rec = rec;
        w_sib = wd.sib;
    } else if (tid == 0) {
        uint32_t r[4];
        Philox ph(f.seed);
        ph(f.offset + (f.rng ? f.rng[0] : 0ull) + (uint64_t)b, 0x554e4946ull, r);
        idx = (int64_t)__umul64hi(((uint64_t)r[0] << 32) | r[1], (uint64_t)f.size);
        s_i64[0] = idx;
    }
    PRISM_STAMP(29);
    // The sampled slot is known.  ONE round of loads: the first O / 4 threads their piece of the CURRENT observation, and
    // every one of the first four waves (each for itself: no hand-off) reward / link / flags of the slot -- hop 0 of the
    // n-step walk (timestep_buffer.py:198-238).  Interleaved environment streams link slot i to i + d with one d for the
    // whole chain, so the link of hop 0 predicts every later hop: the SECOND round asks for reward / link / flags of the
    // slots idx + 2d, idx + 3d, ... (hop k in lane k - 1 of wave 0) and, O / 4 threads again, for the successor
    // observation of the slot the walk should end at.  The walk then runs out of registers and checks every link it
    // follows against the slot that was asked for (a miss falls back to a dependent load); a walk that ends elsewhere
    // (episode end: the successor is the observation itself; a truncation or an open chain) takes the dependent row load
    // further down.  Two dependent round trips where the plain walk made n_step + 1; same values, same order.
    __syncthreads();
    PRISM_STAMP2(2);
    const int O = rp.obs_elems;     // == 100 * C
    const int64_t idx0 = s_i64[0];
    const float *src_obs = rp.obs + idx0 * O;
    float *d0 = f.obs + (int64_t)b * O, *d1 = f.next_obs + (int64_t)b * O;
    const int n_step = rp.n_step;
    float4 xc = make_float4(0.f, 0.f, 0.f, 0.f), xs = xc;
    float crw = 0.f;
    int32_t cnx = -1, act0 = 0;
    uint32_t cfl = 0;
    int64_t pred = -1;
    bool pred_ok = false;
    float rw0 = 0.f;
    int32_t nx0 = -1;
    uint32_t f0 = 0;
    int64_t dlt = 0;
    double g_lane = 0.0;
    {
        if (tid < O / 4) xc = reinterpret_cast<const float4 *>(src_obs)[tid];
        {   // gamma ** lane, read from the argument segment with a per-lane index (indexing the by-value struct at run time
            // would move it to scratch; one scalar load per hop put ~200 cycles into every step of the walk)
            typedef const double __attribute__((address_space(4))) *cdp;
            constexpr size_t RP_OFF = (sizeof(IqnArgs) + alignof(prism_replay_desc) - 1) / alignof(prism_replay_desc) * alignof(prism_replay_desc);
            cdp g = (cdp)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + RP_OFF +
                          offsetof(prism_replay_desc, gammas));
            const int gi = (tid & 63) < n_step ? (tid & 63) : n_step;
            g_lane = g[gi];
        }
        rw0 = rp.reward[idx0];
        nx0 = rp.link[idx0];
        f0 = rp.flags[idx0];
        if (tid == 0) act0 = rp.action[idx0];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PRISM_STAMP2(3);
        const bool go = (f0 & PRISM_FLAG_HAS_NEXT) && !(f0 & PRISM_FLAG_TRUNC) && n_step > 1 && nx0 >= 0;
        dlt = (int64_t)nx0 - idx0;
        pred = go ? idx0 + (int64_t)(n_step - 1) * dlt : idx0;
        pred_ok = pred >= 0 && pred < rp.capacity;
        if (tid < O / 4) {
            obs_to_lds(s_obs[0], xc, tid, C);
            if (pred_ok) xs = reinterpret_cast<const float4 *>(rp.succ_obs + pred * O)[tid];
        }
        if (go && tid < n_step - 1) {                 // (wave 0: n_step <= PRISM_MAX_NSTEP < 64)
            const int64_t c = (int64_t)nx0 + (int64_t)tid * dlt;
            if (c >= 0 && c < rp.capacity) {
                crw = rp.reward[c];
                cnx = rp.link[c];
                cfl = rp.flags[c];
            }
        }
        if (tid < O / 4) stream_store4(reinterpret_cast<float4 *>(d0) + tid, xc);
        if (f.use_per && tid < 64) {
            if (tid < 63 - __clzll((unsigned long long)rp.tree_capacity)) s_sibrec[tid] = w_sib;
            if (tid == 0) {
                s_out_w = pow_neg_beta(w_leaf / w_pmin, f.beta);
                s_rec_state = w_rec;
            }
        }
    }
    lds_barrier();
    PRISM_STAMP2(4);
    conv_embed_rows(s_obs[0], s_w[0], s_b[0], C, a.ws.e_cur + (int64_t)b * E_DIM, tid);
    PRISM_STAMP2(5);
    if (tid < 64) {
        // the walk, wave-uniform (nstep_walk's arithmetic and order)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PRISM_STAMP2(6);
        auto gam = [&](int k) {
            const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(g_lane) & 0xffffffffll), k);
            const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(g_lane) >> 32), k);
            return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        };
        int64_t cur = idx0;
        double ret = (double)rw0 * gam(0);
        uint32_t fl = f0;
        int32_t nx = nx0;
        bool go = (f0 & PRISM_FLAG_HAS_NEXT) && !(f0 & PRISM_FLAG_TRUNC) && n_step > 1 && nx0 >= 0;
        int k = 1;
        for (; go; ++k) {
            cur = nx;
            float rw;
            if (cur == (int64_t)nx0 + (int64_t)(k - 1) * dlt) {
                rw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(crw), k - 1));
                nx = __builtin_amdgcn_readlane(cnx, k - 1);
                fl = (uint32_t)__builtin_amdgcn_readlane((int)cfl, k - 1);
            } else {
                rw = rp.reward[cur];
                nx = rp.link[cur];
                fl = rp.flags[cur];
            }
            ret += (double)rw * gam(k);
            go = (fl & PRISM_FLAG_HAS_NEXT) && !(fl & PRISM_FLAG_TRUNC) && k != n_step - 1 && nx >= 0;
        }
        const double gamma = gam(k);          // gamma ** (number of rewards summed)
        if (tid == 0) {
            s_i64[1] = cur;
            s_flags = fl;
            // 1 = the row that was asked for is the successor; 2 = no successor stored: the observation itself; 0 = fetch it
            s_rec_state2 = !(fl & PRISM_FLAG_HAS_NEXT) ? 2u : (pred_ok && cur == pred) ? 1u : 0u;
            f.reward[b] = (float)ret;
            f.gamma[b] = (float)gamma;
            f.nonterminal[b] = (fl & PRISM_FLAG_DONE) ? 0 : 1;
            f.action[b] = (int64_t)act0;
            f.out_index[b] = idx0;
            if (f.use_per) {
                f.out_weight[b] = s_out_w;
                if (b == 0 || s_rec_state == 2u) atomicMax(a.ws.ticket + 3, s_rec_state);
            }
        }
    } else if (f.use_per && tid - 64 < 63 - __clzll((unsigned long long)rp.tree_capacity)) {
        // the sibling record goes out from lanes that have nothing else to do (level s from lane 64 + s)
        reinterpret_cast<float2 *>(a.ws.sib)[(int64_t)(tid - 64) * B + b] = s_sibrec[tid - 64];
    }
    if (tid < O / 4) obs_to_lds(s_obs[1], xs, tid, C);
    PRISM_STAMP(30);
    lds_barrier();
    const unsigned int spec = __builtin_amdgcn_readfirstlane(s_rec_state2);
    if (spec != 1u) {
        if (spec == 0u && tid < O / 4) xs = reinterpret_cast<const float4 *>(rp.succ_obs + s_i64[1] * O)[tid];
        if (spec == 2u) xs = xc;
        if (tid < O / 4) obs_to_lds(s_obs[1], xs, tid, C);
        lds_barrier();
    }
    if (tid < O / 4) stream_store4(reinterpret_cast<float4 *>(d1) + tid, xs);
    conv_embed_rows(s_obs[1], s_w[1], s_b[1], C, a.ws.e_next + (int64_t)b * E_DIM, tid);
    PRISM_STAMP(31);
}

// ------------------------------------------------------------------------------------------
// post: 1024-thread blocks, three roles.
//   [0, POST_SLAB_BLOCKS)                 slab sum -> grads[phi_w .. w1]
//   [.., + n_conv)                        conv backward partials; the block that finishes last adds them up
//   [.., + POST_SMALL_BLOCKS)             small tensors (b1, LN2, W2, b2) by 64-wide slices of H, + total loss
// Every block leaves its sum of squares in normpart[blockIdx.x].
// ------------------------------------------------------------------------------------------
// one thread per float4 of the slab; two (adjacent lanes, eight chunks each) when the backward wrote sixteen row chunks
__host__ __device__ inline int post_slab_blocks(int slab, int n_chunks = 8) { return ((slab / 4) * (n_chunks > 8 ? 2 : 1) + 1023) / 1024; }
__host__ __device__ inline int post_small_blocks(int H) { return H / SMALL_W; }     // 16 hidden units per block
// Q-head slabs (one per head: no row chunks to add up, only the Theil term and the norm partial): a float4 per thread and
// trip, at most POST_Q_SLAB_MAX workgroups -- ten heads were 326 workgroups of one trip, which alone kept the launch (640
// workgroups, 2.5 rounds of the CUs) from being resident at once, i.e. from carrying the fused tail
constexpr int POST_Q_SLAB_MAX = 64;
__host__ __device__ inline int post_q_slab_blocks(int n_heads, int q_slab) {
    const int full = (n_heads * (q_slab / 4) + 1023) / 1024;
    return full < POST_Q_SLAB_MAX ? full : POST_Q_SLAB_MAX;
}
__host__ __device__ inline int post_conv_blocks(int B, int C) { return (B + conv_spb(C) - 1) / conv_spb(C); }
__host__ __device__ inline int post_blocks(int B, int C, int use_iqn, int n_heads, bool conv_in_bwd, int slab, int q_slab, int Hi,
                                           int Hq, int n_chunks = 8) {
    int n = conv_in_bwd ? 1 : post_conv_blocks(B, C);
    if (use_iqn) n += post_slab_blocks(slab, n_chunks) + post_small_blocks(Hi);
    if (n_heads) n += post_q_slab_blocks(n_heads, q_slab) + n_heads * post_small_blocks(Hq);
    return n;
}
// one-layer DQN head: the loss kernel leaves one conv partial row per sample; CONV_FOLD_W outputs per fold block
constexpr int CONV_FOLD_W = 64;
__host__ __device__ inline int dqn1_conv_blocks(int C) { return (16 * 9 * C + 16 + CONV_FOLD_W - 1) / CONV_FOLD_W; }
__host__ __device__ inline int post_blocks_dqn1(int C) { return dqn1_conv_blocks(C) + DQN_GRAD_BLOCKS; }

__device__ __forceinline__ float block_sum_1024(float v, float *s_red) {
    const int tid = threadIdx.x;
    v = wave_sum(v);
    lds_barrier();              // (LDS only: a __syncthreads() here also waits for every store the caller has in flight)
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    lds_barrier();
    float t = 0.f;
    if (tid < 16) t = s_red[tid];
    if (tid < 64) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    }
    return t;   // valid in thread 0
}

// ---- global-norm clip + Adam over the flat buffers ---------------------------------------------
struct AdamArgs {
    float *p;
    const float *g;
    float *m, *v;
    int64_t n;
    int64_t *step;
    const float *normpart;
    int n_slots;
    double lr, b1, b2, eps;
    float max_norm, grad_scale;
    float *out_scalars;
    unsigned int *ticket;
    const unsigned int *poison;    // data parallel: the workspace status word (PRISM_WS_STATUS_COLLECTIVE_TIMEOUT), else NULL
};

// One NT-thread block of the clip + Adam update (block `blk` of `nblk`).  The operands of the block's first
// float4 per thread are requested BEFORE the norm is folded: the fold's own loads and two barriers then ride on the
// same memory round trip.  The fold itself is always the 256-lane form (strided partial sums, LDS tree), whatever NT:
// every launch shape arrives at the same bits for the norm.
template <int NT>
__device__ __forceinline__ void clip_adam_block(const AdamArgs &a, int blk, int nblk) {
    __shared__ float s_red[256];
    __shared__ float s_c[4];     // clip coef, -step_size, sqrt(bias_correction2)
    const int tid = threadIdx.x;
    // the all-reduce in front of this launch gave up on a peer (direct.hip): the gradient is not a sum over all ranks --
    // apply NOTHING (uniform over the grid: every block reads the same sticky word; the host raises at its next poll)
    if (a.poison && (__hip_atomic_load(a.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & PRISM_WS_STATUS_COLLECTIVE_TIMEOUT)) return;
    const int64_t nvec = a.n >> 2;
    const int64_t i0 = (int64_t)blk * NT + tid;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), p0 = g0, m0 = g0, v0 = g0;
    if (i0 < nvec) {
        g0 = reinterpret_cast<const float4 *>(a.g)[i0];
        p0 = reinterpret_cast<const float4 *>(a.p)[i0];
        m0 = reinterpret_cast<const float4 *>(a.m)[i0];
        v0 = reinterpret_cast<const float4 *>(a.v)[i0];
    }
    // every block folds the same partials in the same order -> identical norm everywhere
    if (tid < 256) {
        float s = 0.f;
#pragma unroll 4
        for (int i = tid; i < a.n_slots; i += 256) s += a.normpart[i];
        s_red[tid] = s;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        const float total = sqrtf(s_red[0]);
        float coef = a.max_norm / (total + 1e-6f);   // torch.nn.utils.clip_grad_norm_
        coef = fminf(coef, 1.0f);
        // torch.optim.Adam (_single_tensor_adam): bias corrections in float64 from the step count
        const double t = (double)(a.step[0] + 1);
        const double bc1 = 1.0 - pow(a.b1, t), bc2 = 1.0 - pow(a.b2, t);
        s_c[0] = coef;
        s_c[1] = (float)(-(a.lr / bc1));
        s_c[2] = (float)sqrt(bc2);
        if (blk == 0) {
            a.out_scalars[3] = total;
            a.out_scalars[5] = coef;
        }
    }
    __syncthreads();
    const float coef = s_c[0], neg_step = s_c[1], bc2s = s_c[2];
    const float w1 = (float)(1.0 - a.b1), b2f = (float)a.b2, w2 = (float)(1.0 - a.b2), epsf = (float)a.eps;
    const float gs = a.grad_scale;
    auto upd = [&](float g_, float &p, float &m, float &v) {
        const float g = (g_ * gs) * coef;
        m = fmaf(w1, g - m, m);                 // exp_avg.lerp_(grad, 1 - beta1)
        v = v * b2f + (w2 * g) * g;             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
        const float denom = sqrtf(v) / bc2s + epsf;
        p = p + (neg_step * m) / denom;         // param.addcdiv_(exp_avg, denom, value=-step_size)
    };
    auto upd4 = [&](int64_t i, const float4 &g, float4 p, float4 m, float4 v) {
        upd(g.x, p.x, m.x, v.x);
        upd(g.y, p.y, m.y, v.y);
        upd(g.z, p.z, m.z, v.z);
        upd(g.w, p.w, m.w, v.w);
        stream_store4(reinterpret_cast<float4 *>(a.p) + i, p);
        stream_store4(reinterpret_cast<float4 *>(a.m) + i, m);
        stream_store4(reinterpret_cast<float4 *>(a.v) + i, v);
    };
    if (i0 < nvec) upd4(i0, g0, p0, m0, v0);
    for (int64_t i = i0 + (int64_t)nblk * NT; i < nvec; i += (int64_t)nblk * NT)
        upd4(i, reinterpret_cast<const float4 *>(a.g)[i], reinterpret_cast<float4 *>(a.p)[i],
             reinterpret_cast<float4 *>(a.m)[i], reinterpret_cast<float4 *>(a.v)[i]);
    if (blk == 0 && tid < (int)(a.n & 3)) {
        const int64_t i = (nvec << 2) + tid;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        upd(a.g[i], p, m, v);
        a.p[i] = p;
        a.m[i] = m;
        a.v[i] = v;
    }
    // the block that finishes last advances the step counter (every block has read it by then)
    __syncthreads();
    if (tid == 0) {
        const unsigned int done = atomicAdd(a.ticket, 1u);
        if (done == (unsigned)(nblk - 1)) {
            a.step[0] = a.step[0] + 1;
            *a.ticket = 0u;
        }
    }
}

// Grid-wide barrier for a launch whose workgroups are ALL resident at once (the host checks the occupancy of the very
// instantiation it launches, per device, before it picks a kernel that calls this), in two halves so that the caller can
// put loads between them.  *bar counts arrivals over ALL launches and is never reset (64-bit: never wraps).  Every launch
// adds exactly GRID_EPOCH to it WHATEVER ITS SHAPE: workgroup 0 weighs GRID_EPOCH - (n - 1), every other workgroup 1 -- so
// a workgroup that drew `old` belongs to launch old >> 20, which is through when the count reaches the next multiple of
// GRID_EPOCH: no second atomic, no reset, no generation word, and launches of different shapes (writeback riding along or
// not, another batch size) may share one workspace without desynchronising the phase.
//   arrive: every wave drains its own stores, the workgroup meets, one lane writes the L2 back (release) and draws
//           its ticket (the atomic is only ISSUED here);
//   wait:   unless it was the last to arrive, that lane polls the counter past the L2 (agent scope) -- for at most
//           GRID_WAIT_TICKS of the 100 MHz real-time counter: a grid that is NOT resident at once after all (another
//           process took part of the device) must not spin for ever.  A workgroup that gives up sets bit 0 of *status
//           (sticky; HipAgent.check_status raises on it) and skips its update.
// There is NO acquire: what a workgroup reads of the others' data behind the barrier it must read with agent-scope
// loads (they do not stop at this XCD's L2) -- one cache invalidation less on everybody's path.
constexpr unsigned long long GRID_EPOCH = 1ull << 20;
constexpr unsigned long long GRID_WAIT_TICKS = 10000000ull;       // 100 ms; a healthy barrier takes about 2 us
constexpr unsigned int GRID_STATUS_TIMEOUT = 1u;
__device__ __forceinline__ unsigned long long grid_barrier_weight(unsigned int n) {
    return blockIdx.x == 0 ? GRID_EPOCH - (unsigned long long)(n - 1) : 1ull;
}
// `how` = 2, light: the workgroup hands NOTHING to the others through ordinary stores (its only cross-workgroup word was
// written at agent scope by thread 0, which waits for that one store): no drain of the other waves, no L2 write-back.
// `how` = 1: everything the others read was stored at agent scope (far_store): every wave waits for its own stores, the
// workgroup meets, no L2 write-back.  `how` = 0: the full release.
__device__ __forceinline__ unsigned long long grid_barrier_arrive(unsigned long long *bar, unsigned int n, int how = 0) {
    unsigned long long old = 0;
    if (how == 2) {
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            old = atomicAdd(bar, grid_barrier_weight(n));
        }
        return old;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (how == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        old = atomicAdd(bar, grid_barrier_weight(n));
    }
    return old;
}
// returns false (in every thread) when the wait was abandoned
__device__ __forceinline__ bool grid_barrier_wait(unsigned long long *bar, unsigned long long old, unsigned int n,
                                                  unsigned int *status, unsigned int *host_status = nullptr) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        const unsigned long long target = ((old >> 20) + 1ull) << 20;
        int ok = 1;
        if (old + grid_barrier_weight(n) < target) {
            // (the clock is read once every 256 polls: a read of the real-time counter in EVERY iteration made the waiter
            // notice the last arrival later -- tail launch 14.0 -> 15.2 us)
            unsigned long long t0 = 0;
            unsigned int polls = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++polls & 255u) == 0u) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > GRID_WAIT_TICKS) {
                        ok = 0;
                        atomicOr(status, GRID_STATUS_TIMEOUT);
                        // (word 0 of the host's pinned status words = bit 0: polled every step without a device sync)
                        if (host_status) __hip_atomic_store(host_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

// the clip + Adam half of the step when it rides in the post launch (single GPU: nothing sits between the two)
struct TailArgs {
    AdamArgs adam;
    unsigned long long *barrier;   // [1] arrivals, cumulative over all launches
    unsigned int *status;          // [1] sticky GRID_STATUS_* bits
    unsigned int *host_status;     // optional pinned host words [4] mirroring the sticky bits (prism_learner_desc.host_status)
    uint64_t *rng;             // device RNG counters {PER draws, tau draws} advanced once per step (or NULL)
    uint64_t inc_per, inc_tau;
};

// Which float4 of the flat vectors a thread of the fused tail updates.  A slab block keeps the gradient it has just
// summed (`own`: no reload); the other role blocks share what lies outside the slab range [s0, s0 + cnt): leftover
// number k is float4 k below s0 and k + cnt above.
struct TailShare {
    int64_t j;        // first float4 of this thread
    bool have, own;
    float4 g;         // the owned gradient
    int64_t k, kstride, nleft, s0, cnt;     // leftover walk (kstride 0: none)
    bool scalar_tail; // this block also updates the n & 3 trailing elements
    float *gout;      // slab blocks: the flat gradient, to which the owned sums are written behind the arrival; else NULL
    int arrive;       // 0 full release, 1 drain only (every cross-read store was written through), 2 light (see grid_barrier_arrive)
};

// Clip + Adam behind the grid barrier (1024 threads).  Everything that does not depend on the other workgroups is
// requested or computed between the barrier's two halves and lands while the workgroup waits: the thread's parameter /
// moment vectors (issued AFTER the arrival ticket -- memory operations of a wave return in order, a ticket queued behind
// them would wait for them) and the float64 bias corrections (another wave than the one that polls).  Behind it: the norm partials (and
// the gradient where it is not owned), the 256-lane fold every launch shape uses, the update.  Bit-identical to
// clip_adam_block.
__device__ __forceinline__ void tail_clip_adam(const AdamArgs &a, unsigned long long *barrier, unsigned int *status, int n_role,
                                               const TailShare &sh, int64_t step_now, unsigned long long *st,
                                               unsigned int *host_status = nullptr) {
    auto stamp = [&](int k) {
        if (st && threadIdx.x == 0) {
            st[(size_t)blockIdx.x * 64 + k] = __builtin_amdgcn_s_memtime();
            st[(size_t)blockIdx.x * 64 + 32 + k] = __builtin_amdgcn_s_memrealtime();
        }
    };
    __shared__ float s_red[256];
    __shared__ float s_c[4];
    const int tid = threadIdx.x;
    // (the step count was requested at kernel entry; it has to have ARRIVED before this workgroup is counted in --
    // workgroup 0 overwrites it behind the barrier)
    asm volatile("" ::"s"((int)step_now), "s"((int)(step_now >> 32)));
    const unsigned long long ticket = grid_barrier_arrive(barrier, (unsigned)n_role, sh.arrive);
    if (sh.gout && sh.own) stream_store4(reinterpret_cast<float4 *>(sh.gout) + sh.j, sh.g);
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), m0 = p0, v0 = p0;
    if (sh.have) {
        p0 = reinterpret_cast<const float4 *>(a.p)[sh.j];
        m0 = reinterpret_cast<const float4 *>(a.m)[sh.j];
        v0 = reinterpret_cast<const float4 *>(a.v)[sh.j];
    }
    if (tid == 64) {
        const double t = (double)(step_now + 1);
        const double bc1 = 1.0 - pow(a.b1, t), bc2 = 1.0 - pow(a.b2, t);
        s_c[1] = (float)(-(a.lr / bc1));
        s_c[2] = (float)sqrt(bc2);
    }
    if (!grid_barrier_wait(barrier, ticket, (unsigned)n_role, status, host_status)) return;      // (uniform: abandoned, flagged)
    stamp(25);
    // (everything another workgroup wrote is read at agent scope: the barrier has no acquire)
    auto far = [](const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto far4 = [&](int64_t i) { return make_float4(far(a.g + 4 * i), far(a.g + 4 * i + 1), far(a.g + 4 * i + 2), far(a.g + 4 * i + 3)); };
    float4 g0 = sh.g;
    if (sh.have && !sh.own) g0 = far4(sh.j);
    if (tid < 256) {
        float s = 0.f;
#pragma unroll 4
        for (int i = tid; i < a.n_slots; i += 256) s += far(a.normpart + i);
        s_red[tid] = s;
    }
    if (blockIdx.x == 0 && tid == 64) a.step[0] = step_now + 1;        // (every workgroup read it before it arrived)
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        const float total = sqrtf(s_red[0]);
        float coef = a.max_norm / (total + 1e-6f);
        coef = fminf(coef, 1.0f);
        s_c[0] = coef;
        if (blockIdx.x == 0) {
            a.out_scalars[3] = total;
            a.out_scalars[5] = coef;
        }
    }
    __syncthreads();
    stamp(15);
    const float coef = s_c[0], neg_step = s_c[1], bc2s = s_c[2];
    const float w1 = (float)(1.0 - a.b1), b2f = (float)a.b2, w2 = (float)(1.0 - a.b2), epsf = (float)a.eps;
    const float gs = a.grad_scale;
    auto upd = [&](float g_, float &p, float &m, float &v) {
        const float g = (g_ * gs) * coef;
        m = fmaf(w1, g - m, m);
        v = v * b2f + (w2 * g) * g;
        const float denom = sqrtf(v) / bc2s + epsf;
        p = p + (neg_step * m) / denom;
    };
    auto upd4 = [&](int64_t i, const float4 &g, float4 p, float4 m, float4 v) {
        upd(g.x, p.x, m.x, v.x);
        upd(g.y, p.y, m.y, v.y);
        upd(g.z, p.z, m.z, v.z);
        upd(g.w, p.w, m.w, v.w);
        stream_store4(reinterpret_cast<float4 *>(a.p) + i, p);
        stream_store4(reinterpret_cast<float4 *>(a.m) + i, m);
        stream_store4(reinterpret_cast<float4 *>(a.v) + i, v);
    };
    if (sh.have) upd4(sh.j, g0, p0, m0, v0);
    if (sh.kstride)
        for (int64_t k = sh.k + sh.kstride; k < sh.nleft; k += sh.kstride) {
            const int64_t i = k < sh.s0 ? k : k + sh.cnt;
            upd4(i, far4(i), reinterpret_cast<float4 *>(a.p)[i], reinterpret_cast<float4 *>(a.m)[i],
                 reinterpret_cast<float4 *>(a.v)[i]);
        }
    if (sh.scalar_tail && tid < (int)(a.n & 3)) {
        const int64_t i = ((a.n >> 2) << 2) + tid;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        upd(far(a.g + i), p, m, v);
        a.p[i] = p;
        a.m[i] = m;
        a.v[i] = v;
    }
}

struct PostWriteback {        // optional: prism_per_update(index, |out_td|) as the last block of the post launch
    prism_replay_desc rp;
    const int64_t *index;
    const float2 *sib;        // sibling record of the front kernel's descents ([level][B])
    unsigned int *sib_state;  // 0 = none, 1 = valid for `index`, 2 = unusable (a sample was clamped); consumed by
                              // whoever runs the level walk
    int4 *plan;               // non-NULL: only prepare here (ranking, winners); step_back_kernel finishes
    float alpha, eps;
    int enabled, block;
};

// WB_FULL: the writeback block runs the whole update (batches above 256); otherwise it only prepares
// it (two instantiations: the full writer's register arrays would otherwise tax every role with spills)
// TAIL: the launch also clips and applies Adam (single GPU): a grid barrier after the partial norms, then every role
// block updates its share of the flat parameter vector; the writeback block (always the full writer then) advances the
// device RNG counters and does not take part in the barrier.
template <bool WB_FULL, bool TAIL, bool DENSE>
__global__ __launch_bounds__(1024) void iqn_post_kernel(IqnArgs a, PostWriteback wb, TailArgs tl) {
    static_assert(WB_FULL || !TAIL, "the fused tail has nobody to finish a prepared writeback");
    kernarg_prefetch<sizeof(IqnArgs) + sizeof(PostWriteback) + sizeof(TailArgs)>();
    // (fused tail) the optimizer's step count, read before anything else: workgroup 0 advances it behind the barrier
    const int64_t step_now = TAIL ? tl.adam.step[0] : 0;
    // conv-backward staging and the writeback scratch never coexist in one block: one aliased pool
    constexpr int POOL = PER_UPDATE_LDS_BYTES > (int)(CONV_LDS_FLOATS * sizeof(float)) ? PER_UPDATE_LDS_BYTES
                                                                                       : (int)(CONV_LDS_FLOATS * sizeof(float));
    // (one role per workgroup: the roles never share the pool in time; each must fit it)
    static_assert(POOL >= (int)(SMALL_POOL_FLOATS * sizeof(float)), "small-tensor fold must fit the pool");
    static_assert(POOL >= (int)(16 * CONV_FOLD_W * sizeof(float)) && POOL >= (int)(1024 * sizeof(float)),
                  "conv fold staging must fit the pool");
    __shared__ __attribute__((aligned(16))) char s_pool[POOL];
    __shared__ float s_red[64];
    PRISM_STAMP(13);
    // Which role: the launch's workgroups start in index order and, when they are not all resident at once (c4: 640 of
    // them, 2.5 rounds of the CUs), the ones that start last had better be short.  The longest roles sit at the END of the
    // role order (the Q heads' small-tensor folds, the priority writeback): the split form rotates them to the front.
    // (The fused tail keeps the identity: its launch is resident at once, and its share / rank arithmetic is by index.)
    int bid = blockIdx.x;
    if constexpr (!TAIL) {
        const int rot = (a.head_layers == 2 ? a.n_heads * post_small_blocks(a.Hq) : 0) + (wb.enabled ? 1 : 0);
        const int nb = (int)gridDim.x;
        if (rot < nb) bid = bid < rot ? nb - rot + bid : bid - rot;
    }
    if (wb.enabled && bid == wb.block) {
        // (models with both parts: td = dl / 2 + ql / 2 is combined HERE from the two losses -- out_td is written by another
        // block of this launch, the Q loss no longer reads the IQN loss's output: the two loss kernels are one launch)
        const bool both = a.use_iqn && a.n_heads > 0;
        const float *pr = both ? a.out_dl : a.out_td, *pr2 = both ? a.out_ql : nullptr;
        if (!WB_FULL) {
            per_update_block<true>(wb.rp, wb.index, pr, a.B, wb.alpha, wb.eps, 1, s_pool, nullptr, 0, wb.plan, 0, pr2);
        } else {
            const unsigned int rec = wb.sib_state ? *wb.sib_state : 0u;
            // waves the batch does not need leave now (one pass: B <= 512)
            const int live = a.B <= UPD_MAX ? min(1024, 2 * ((a.B + 63) & ~63)) : 0;     // (x2: the ranking splits its count two ways)
            if (live && (int)threadIdx.x >= live) return;
            per_update_block<false, DENSE>(wb.rp, wb.index, pr, a.B, wb.alpha, wb.eps, 1, s_pool,
                                           rec == 1u ? wb.sib : nullptr, a.B, nullptr, live, pr2);
            if (wb.sib_state && threadIdx.x == 0) *wb.sib_state = 0u;  // (every thread read it before its first barrier)
        }
        if (TAIL && tl.rng && threadIdx.x == 0) {
            tl.rng[0] += tl.inc_per;
            tl.rng[1] += tl.inc_tau;
        }
        PRISM_STAMP(14);
        return;
    }
    __shared__ float s_kappa[Q_MAX_HEADS];
    __shared__ float s_parts[Q_MAX_HEADS * Q_NORM_PARTS], s_norm2[Q_MAX_HEADS];
    const int tid = threadIdx.x, B = a.B, C = a.C;
    const bool dqn1 = a.head_layers == 1 && a.n_heads;
    const int n_conv = dqn1 ? dqn1_conv_blocks(C) : (a.conv_in_bwd ? 1 : post_conv_blocks(B, C));
    int blk = bid;
    float sq = 0.f;
    float4 g_own = make_float4(0.f, 0.f, 0.f, 0.f);     // (fused tail) the slab sum of this thread, kept for its Adam update
    bool far_all = false;       // every store of this role that another workgroup reads behind the barrier was a far_store
    int norm_slot = bid;        // where this workgroup's sum of squared gradients goes (-1: it has none to report)
    if (dqn1 && blk < n_conv) {
        // fold the per-sample rows of the loss kernel: 64 outputs x 16 batch parts per workgroup
        float *s_part = reinterpret_cast<float *>(s_pool);           // [16][CONV_FOLD_W]
        const int nk = 9 * C, n_out = 16 * nk + 16;
        const int ol = tid & (CONV_FOLD_W - 1), part = tid >> 6, o = blk * CONV_FOLD_W + ol;
        float t = 0.f;
        if (o < n_out) {
#pragma unroll 16
            for (int b = part; b < B; b += 16) t += a.ws.convpart[(int64_t)b * CONV_ROW + o];
        }
        s_part[part * CONV_FOLD_W + ol] = t;
        __syncthreads();
        if (part == 0 && o < n_out) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += s_part[q * CONV_FOLD_W + ol];
            a.grads[(o < 16 * nk ? a.off.conv_w : a.off.conv_b - 16 * nk) + o] = v;
            sq += v * v;
        }
    } else if (a.conv_in_bwd && blk == 0) {
        // the backward kernel left one partial row per (row chunk rc, column slice cs): channel c owns
        // slices 4c..4c+3.  Fold them in fixed order (rc outer, slice inner).
        const int nk = 9 * C, n_out = 16 * nk + 16, n_cs = E_DIM / 16;
        PRISM_STAMP(20);
        // every partial requested before the first add (one row chunk per trip of a rolled loop was sixteen dependent
        // round trips).  a.conv_rows: partial rows per (row chunk, channel) -- the 64-column backward adds its four column
        // slices itself (1), the 16-column one leaves them to this fold (4): 64 scalar loads in each of ten waves were
        // 10 k cycles of this one CU's address unit, and the whole grid waits for this workgroup
        for (int o = tid; o < n_out; o += 1024) {
            const int c = o < 16 * nk ? o / nk : o - 16 * nk, k = o < 16 * nk ? o - c * nk : nk;
            const float *src = a.ws.convpart + (int64_t)(4 * c) * BWD_CONV_ROW + k;
            float t = 0.f;
            if (a.conv_rows == 1) {
                float v[16];
#pragma unroll
                for (int rc = 0; rc < 16; ++rc) v[rc] = src[(int64_t)(rc < a.n_chunks ? rc : 0) * n_cs * BWD_CONV_ROW];
#pragma unroll
                for (int rc = 0; rc < 16; ++rc)
                    if (rc < a.n_chunks) t += v[rc];
            } else {
                float v[16][4];
#pragma unroll
                for (int rc = 0; rc < 16; ++rc) {
                    const float *r = src + (int64_t)(rc < a.n_chunks ? rc : 0) * n_cs * BWD_CONV_ROW;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[rc][q] = r[q * BWD_CONV_ROW];
                }
#pragma unroll
                for (int rc = 0; rc < 16; ++rc)
                    if (rc < a.n_chunks) t += ((v[rc][0] + v[rc][1]) + v[rc][2]) + v[rc][3];
            }
            far_store(&a.grads[(o < 16 * nk ? a.off.conv_w : a.off.conv_b - 16 * nk) + o], t);
            sq += t * t;
        }
        far_all = true;
        PRISM_STAMP(21);
    } else if (blk < n_conv) {
        __shared__ int s_last;
        conv_bwd_partial_block(a, blk, reinterpret_cast<float *>(s_pool));
        PRISM_STAMP(20);
        // publish, then let the last arriver fold all partial rows (the placement-independent hand-off of the
        // CDNA guide): EVERY storing wave drains its stores (a barrier alone only proves they were issued),
        // the workgroup barrier, then ONE lane releases (L2 write-back), waits again (hipcc may drop the fence's
        // own wait) and draws the ticket; the last arriver acquires before any of its waves reads
        // (round 4: the rows are written THROUGH (far_store) and the last arriver reads them at agent scope: no release fence
        // -- an L2 write-back: 3.6 us between "partial rows out" and the ticket by the stamps of the additive ablation preset,
        // whose fused tail waits for exactly this role -- and no acquire on the other side)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_last = (atomicAdd(a.ws.ticket + 1, 1u) == (unsigned)(n_conv - 1));
        __syncthreads();
        PRISM_STAMP(21);
        // The conv gradient's squares are summed by WHICHEVER workgroup arrived last: reported in that workgroup's own slot they
        // moved from slot to slot between runs, and with them the order in which the partial norms are added -- the global norm
        // (and, once it clips, every parameter) differed in the last bit from one run of the same seed to the next (found by
        // tools/soak.py twin: two learners of one seed side by side).  They always go to the role's FIRST slot, which nobody
        // else writes; every other conv slot gets its workgroup's zero.
        norm_slot = s_last ? 0 : (bid == 0 ? -1 : bid);
        if (s_last && bid != 0 && tid == 0) __hip_atomic_store(a.ws.normpart + bid, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_last) {
            // fold all partial rows: item = (output, half of the rows), combined through LDS
            const int nk = 9 * C, n_out = 16 * nk + 16;
            float *s_half = reinterpret_cast<float *>(s_pool);
            const int hrows = (n_conv + 1) / 2;
            if (n_conv <= 16) {
                // few rows (batch 64: eight): one thread an output, every row requested at once -- one trip, no LDS, no barrier
                // (as (output, half) items the 592 outputs of C = 4 took two passes of 512: two trips and four barriers on the
                // path the additive presets' fused tail waits for)
                for (int o = tid; o < n_out; o += 1024) {               // (up to 1456 outputs at ten channels: two trips there)
                    float v[16];
#pragma unroll
                    for (int ch = 0; ch < 16; ++ch) v[ch] = far_load(&a.ws.convpart[(int64_t)min(ch, n_conv - 1) * CONV_ROW + o]);
                    float t = 0.f;
#pragma unroll
                    for (int ch = 0; ch < 16; ++ch)
                        if (ch < n_conv) t += v[ch];
                    if (o < 16 * nk) far_store(&a.grads[a.off.conv_w + o], t);
                    else far_store(&a.grads[a.off.conv_b + (o - 16 * nk)], t);
                    sq += t * t;
                }
            } else
            for (int base = 0; base < n_out; base += 512) {
                const int o = base + (tid & 511), half = tid >> 9;
                float s = 0.f;
                if (o < n_out) {
                    const int r0 = half * hrows, r1 = min(n_conv, r0 + hrows);
#pragma unroll 16
                    for (int ch = r0; ch < r1; ++ch) s += far_load(&a.ws.convpart[(int64_t)ch * CONV_ROW + o]);
                }
                __syncthreads();
                s_half[tid] = s;
                __syncthreads();
                if (half == 0 && o < n_out) {
                    const float t = s_half[tid] + s_half[tid + 512];
                    if (o < 16 * nk) far_store(&a.grads[a.off.conv_w + o], t);
                    else far_store(&a.grads[a.off.conv_b + (o - 16 * nk)], t);
                    sq += t * t;
                }
            }
            if (tid == 0) a.ws.ticket[1] = 0u;
        }
        far_all = true;        // (every store of this role another workgroup reads was written through)
    } else {
        blk -= n_conv;
        bool done = false;
        if (a.use_iqn) {
            const int n_slab = post_slab_blocks(a.slab, a.n_chunks), n_small = post_small_blocks(a.Hi);
            if (blk < n_slab) {
                // sixteen row chunks (the 64-column backward): a float4 is summed by TWO adjacent lanes, eight chunks each
                // ((c0 + .. + c7) + (c8 + .. + c15), the same in both lanes); the even lane owns the result
                const bool two = a.n_chunks > 8;
                const int t = blk * 1024 + tid, i = two ? t >> 1 : t, half = two ? (t & 1) : 0;
                PRISM_STAMP(8);
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                const bool in = i < a.slab / 4;
                if (in) {
                    // every chunk's partial requested before the first add, then summed in chunk order
                    const int c0 = 8 * half, nc = two ? 8 : a.n_chunks;
                    s = reinterpret_cast<const float4 *>(a.ws.slabs + (int64_t)c0 * a.slab)[i];
                    float4 v[7];
#pragma unroll
                    for (int c = 1; c < 8; ++c)
                        if (c < nc) v[c - 1] = reinterpret_cast<const float4 *>(a.ws.slabs + (int64_t)(c0 + c) * a.slab)[i];
#pragma unroll
                    for (int c = 1; c < 8; ++c)
                        if (c < nc) {
                            s.x += v[c - 1].x; s.y += v[c - 1].y; s.z += v[c - 1].z; s.w += v[c - 1].w;
                        }
                }
                if (two) {
                    s.x += __shfl_xor(s.x, 1, 64);
                    s.y += __shfl_xor(s.y, 1, 64);
                    s.z += __shfl_xor(s.z, 1, 64);
                    s.w += __shfl_xor(s.w, 1, 64);
                }
                if (in && half == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    PRISM_STAMP(7);
                    // (fused tail: the owner keeps the sum in registers for its Adam update and writes the gradient out
                    // BEHIND its barrier arrival -- nobody else reads it, and a store in flight is a store to drain)
                    if constexpr (!TAIL) stream_store4(reinterpret_cast<float4 *>(a.grads + a.off.phi_w + 4 * (int64_t)i), s);
                    sq = (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
                    g_own = s;
                }
                done = true;
            } else if (blk < n_slab + n_small) {
                small_tensor_block(a, blk - n_slab, sq, reinterpret_cast<float *>(s_pool));
                far_all = true;
                done = true;
            } else {
                blk -= n_slab + n_small;
            }
        }
        if (!done && a.head_layers == 1) {
            dqn_grad_block(a, blk, sq, reinterpret_cast<float *>(s_pool));
            done = true;
        }
        if (!done) {
            // Q-head roles.  Theil gradient factor per head: dL/dtheta += kappa_h * theta with
            // kappa_h = -q_w * coef * mean_b(w_b) * c_h   (q_ensemble.py:86-92, agent.py:62-64)
            const float *kappa = nullptr;
            if (a.theil_coef != 0.f) {
                float mw = 0.f;
                if (a.per_weights) {
                    for (int b = tid; b < B; b += 1024) mw += a.per_weights[b];
                } else if (tid == 0) {
                    mw = (float)B;
                }
                stage_head_norms(a, s_parts, s_norm2);
                const float tot = block_sum_1024(mw, s_red);
                if (tid == 0) {
                    float c[Q_MAX_HEADS], theil;
                    theil_factors(s_norm2, a.n_heads, c, theil);
                    const float f = -a.q_w * a.theil_coef * (tot / (float)B);
                    for (int h = 0; h < a.n_heads; ++h) s_kappa[h] = f * c[h];
                }
                __syncthreads();
                kappa = s_kappa;
            }
            const int nqs = post_q_slab_blocks(a.n_heads, a.q_slab), nqsm = post_small_blocks(a.Hq);
            if (blk < nqs) {
                const int64_t n4 = (int64_t)a.n_heads * (a.q_slab / 4);
#pragma unroll 2
                for (int64_t i = (int64_t)blk * 1024 + tid; i < n4; i += (int64_t)nqs * 1024) q_slab_sum(a, i, kappa, sq);
            } else {
                const int x = blk - nqs;
                // total loss (agent.py:58-64): mean(dl*w) + mean(ql*w); operands requested before the fold
                float lw = 0.f, li = 0.f;
                if (x == 0) {
                    for (int b = tid; b < B; b += 1024) {
                        lw += a.ws.q_lossw[b];
                        if (a.use_iqn) {
                            li += a.ws.lossw[b];
                            a.out_td[b] = a.out_dl[b] * 0.5f + a.out_ql[b] * 0.5f;      // td errors (composite_model.py:135-137)
                        }
                    }
                }
                q_small_tensor_block(a, x / nqsm, x % nqsm, kappa, sq,
                                     reinterpret_cast<float *>(s_pool));
                if (x == 0) {
                    const float tq = block_sum_1024(lw, s_red);
                    const float ti = block_sum_1024(li, s_red);
                    if (tid == 0) {
                        a.out_scalars[2] = tq / (float)B;
                        a.out_scalars[0] = ti / (float)B + tq / (float)B;
                    }
                }
            }
        }
    }
    const float t = block_sum_1024(sq, s_red);
    // (agent scope: written through to where the other XCDs' agent-scope loads look for it, whatever release follows)
    if (tid == 0 && norm_slot >= 0) __hip_atomic_store(a.ws.normpart + norm_slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bid == 0 && tid == 0 && a.has_target) a.ws.ticket[2] = 1u;      // the front / embed launch of this update packed the target set
    PRISM_STAMP(14);
    if constexpr (TAIL) {
        const int n_role = wb.enabled ? (int)gridDim.x - 1 : (int)gridDim.x;
        const int n_slab = a.use_iqn ? post_slab_blocks(a.slab, a.n_chunks) : 0;
        const int sb = (int)blockIdx.x - n_conv;
        TailShare sh;
        sh.nleft = (tl.adam.n >> 2) - (a.use_iqn ? a.slab >> 2 : 0);
        sh.s0 = a.use_iqn ? a.off.phi_w >> 2 : 0;
        sh.cnt = a.use_iqn ? a.slab >> 2 : 0;
        sh.g = g_own;
        if (a.use_iqn && sb >= 0 && sb < n_slab) {
            const bool two = a.n_chunks > 8;
            const int64_t t = (int64_t)sb * 1024 + tid, i = two ? t >> 1 : t;
            sh.own = sh.have = i < sh.cnt && (!two || (t & 1) == 0);
            sh.j = sh.s0 + i;
            sh.k = sh.kstride = 0;
            sh.scalar_tail = false;
            sh.gout = a.grads;
            sh.arrive = 2;
        } else {
            const int rank = (int)blockIdx.x < n_conv ? (int)blockIdx.x : (int)blockIdx.x - n_slab;
            sh.own = false;
            sh.gout = nullptr;
            sh.arrive = far_all ? 1 : 0;
            sh.k = (int64_t)rank * 1024 + tid;
            sh.kstride = (int64_t)(n_role - n_slab) * 1024;
            sh.have = sh.k < sh.nleft;
            sh.j = sh.k < sh.s0 ? sh.k : sh.k + sh.cnt;
            sh.scalar_tail = rank == 0;
        }
        tail_clip_adam(tl.adam, tl.barrier, tl.status, n_role, sh, step_now, (a.dbg & 8) ? (unsigned long long *)a.stamps : nullptr,
                       tl.host_status);
        if (!wb.enabled && tl.rng && blockIdx.x == 0 && tid == 0) {
            tl.rng[0] += tl.inc_per;
            tl.rng[1] += tl.inc_tau;
        }
        PRISM_STAMP(9);
    }
}

// ------------------------------------------------------------------------------------------
// back: block 0 = priority writeback (+ RNG counters); blocks [1, 1 + n_adam) = clip + Adam.
// ------------------------------------------------------------------------------------------
struct BackArgs {
    const int64_t *index;
    const float *priority;
    int n;
    float alpha, eps;
    int take_abs, use_per;
    const int4 *plan;          // non-NULL: the post kernel prepared this writeback; finish it here
    const float2 *sib;
    unsigned int *sib_state;
    uint64_t *rng;
    uint64_t inc_per, inc_tau;
};

}  // namespace prism
