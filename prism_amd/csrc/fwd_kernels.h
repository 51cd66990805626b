// Forward tiles of the TD update for gfx950: one 16-row tile per 512-thread workgroup, weights streamed
// ONCE from L2 straight into MFMA operand registers, activations never leave the chip.
//
// Restates /root/reference/prism/agents/models/iqn_model.py:48-93 (+ ffnn_model.py:61-76) and the head
// stack of q_ensemble.py:25-48.  Three kinds of tile share the code:
//   kind 0  IQN rows (sample, tau) of one pass          input = ReLU(phi(cos basis)) * e
//   kind 1  Q-head rows: 16 samples of one ensemble head input = e
//   kind 2  IQN "mixed" tile: for each of its samples the T current-state rows AND the T next-state rows
//           (same online weights: no target network), so the whole loss of those samples -- argmax,
//           n-step target, T x T quantile-Huber tile, dL/dq, head + LayerNorm backward -- finishes inside the
//           workgroup (iqn_model.py:95-201): no second launch, no cross-workgroup hand-off.
//
// Shape of the computation (everything transposed so that the accumulator of one product IS the operand of
// the next; MFMA 16x16x4 fp32: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[i = 4(l>>4)+r][j = l&15]):
//   Y^T[n][m]    = Wphi[n][:] . cos[m][:]          A = Wphi (streamed), B = cos basis (16 registers per lane)
//   x[n][m]      = ReLU(Y^T + bphi[n]) * e[b(m)][n]   (the bias is added ONCE, to the finished product, as a GEMM
//                                                  epilogue does: accumulating onto it rounds at the bias's magnitude
//                                                  sixteen times, which shows as soon as the bias is large)
//   pre^T[h][m] += W1g[h][n] * x[n][m]             A = W1 (streamed, LayerNorm scale folded in), B = x: the D
//                                                  registers of the first product, used as they stand
// Wave w owns embed columns [128w, 128w + 128): eight 16-column steps, each 16 phi MFMAs + 4 * H/16 trunk
// MFMAs, with no barrier and no LDS traffic between them.  LayerNorm(1024) is applied AFTER the GEMM from
// the row sums the phi epilogue accumulates on the side:
//   LN(x) . W1^T = rstd * (x . (g*W1)^T - mean * u) + v,   u = W1 g, v = W1 beta   (front kernel roles)
// which is what removes the row-wide synchronisation between the two products.  Written like that, both the one-pass
// variance (E[x^2] - mean^2) and the difference x . (g*W1)^T - mean * u cancel catastrophically when a row's mean is large
// against its spread.  So every wave SHIFTS its slice of the row by a constant c close to the row's mean -- the first
// element the wave produces of that row -- before it squares, sums and multiplies:
//   d = x - c_w;   x . (g*W1)^T = sum_w [ d . (g*W1)_w^T + c_w u_w ],   u_w = u restricted to the wave's 128 columns
//   pre = rstd * ( sum_w part_w + sum_w (c_w - mean) u_w ) + v
// and the row statistics are combined from per-wave (c_w, sum d, sum d^2) as shifted moments (Chan et al.): every
// quantity that enters a difference is of the size of the row's spread.  One subtraction per element in the stream.  The eight K-slices are
// folded through LDS in wave order (bitwise reproducible), then LayerNorm(H) + head run with one row per
// 32-lane half-wave.
#pragma once
#include "iqn_kernels.h"

namespace prism {

// Waves that stream.  Eight (two per SIMD, eight 16-column steps each).  The kernel also runs with twelve at width 128
// (three per SIMD taking 6 / 5 / 5 of a SIMD's sixteen steps -- set fw_waves to 12: only the first eight go on behind the
// stream, the others store their partial sums and end, hardware barriers only count waves that have not ended), and that
// is SLOWER on MI355X: stream + wait for the slowest wave stays at 34.7 k ticks (8 waves: 34.5 k), the fold reads 12
// partials, kernel 28.7 vs 27.6 us same-box A/B.  With "removing every load saves 3.5 k" this says the streamed phase is
// bound by what one SIMD issues for a step (MFMA 24.6 k + the epilogue VALU between the chains), not by latency a third
// wave could cover.
constexpr int FW_ROW_WAVES = 8;
__host__ __device__ constexpr int fw_waves(int H) { return 8; }
__host__ __device__ constexpr int fw_threads(int H) { return 64 * fw_waves(H); }
constexpr int FW_STEPS = E_DIM / 16;             // 16-column steps of a tile
constexpr int FW_ZS = 17;                        // row stride of the Z tile in LDS
#ifndef FW_SPLIT_RING
#define FW_SPLIT_RING 4                          // trunk operand groups a wave keeps in flight (3 KB each)
#endif
#ifndef FW_KO
#define FW_KO 0                                  // knock-out experiments (tools/fwd_knockouts.sh): 1 no phi save, 2 no operand split, 4 half the weight bytes
#endif
constexpr int FW_CBS = 36;                       // row stride (dwords) of a bf16 cos plane: 32 dwords of pairs + 4 (rows li, li + 8 share banks: 2-way)

// LDS of a forward tile (floats): cos tile | tau + loss scalars | row-stat partials | K-slice partials | Z tile | head
// weight; the loss of a mixed tile reuses the (folded, dead) K-slice partials for its two per-row products.
// Phases: STREAM = prologue + streamed products, ROWS = fold / LayerNorm / head, LOSS = kind-2 tail.
template <int H, int WAVES = 8>
struct FwLds {
    static constexpr int W = WAVES, HP = H + 4;
    static constexpr unsigned STREAM = 1u, ROWS = 2u, LOSS = 4u;
    // [16][CS] cos basis of the tile's rows (fp32 chain) | [3 planes][16 rows][FW_CBS] packed bf16 pieces (bf16 path)
    static constexpr LdsRegion COST{0, 3 * 16 * FW_CBS > 16 * CS ? 3 * 16 * FW_CBS : 16 * CS, STREAM};
    static constexpr LdsRegion ROWF{COST.off + COST.size, 128, LDS_ALWAYS};                 // tau, loss scalars, b2 at [96, 112)
    static constexpr LdsRegion STAT{ROWF.off + ROWF.size, 3 * W * 16, STREAM | ROWS};       // [3][waves][16 rows]: sum d | sum d^2 | shift
    static constexpr LdsRegion PART{STAT.off + STAT.size, W * 16 * HP, STREAM | ROWS};      // [waves][16 rows][HP]
    static constexpr LdsRegion ZT{PART.off + PART.size, 16 * FW_ZS + 16, ROWS | LOSS};      // [16][FW_ZS]
    static constexpr LdsRegion W2S{ZT.off + ZT.size, 16 * H, LDS_ALWAYS};                   // [A][H] head Linear weight
    static constexpr LdsRegion LOSS_S{PART.off, 16 * HP, LOSS};                             // [16][HP] dq * head input
    static constexpr LdsRegion LOSS_P{PART.off + 16 * HP, 16 * HP, LOSS};                   // [16][HP] dpre1
    static constexpr int TOTAL = W2S.off + W2S.size;
    static constexpr LdsRegion ALL[] = {COST, ROWF, STAT, PART, ZT, W2S, LOSS_S, LOSS_P};
    static_assert(lds_layout_ok(ALL, TOTAL), "forward tile: LDS regions live at the same time overlap");
    static_assert(COST.off % 4 == 0 && PART.off % 4 == 0 && W2S.off % 4 == 0 && HP % 4 == 0, "16-byte accessed regions");
};
template <int H, int WAVES = 8>
__host__ __device__ constexpr int fw_lds_floats() { return FwLds<H, WAVES>::TOTAL; }

// sum over the 32-lane half a lane belongs to; every lane of the half receives the total (fixed order)
__device__ __forceinline__ float half_sum(float v) {
    v += dpp_move<0xB1, 0xF>(0.f, v);     // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xF>(0.f, v);     // quad_perm [2,3,0,1]
    v += dpp_move<0x124, 0xF>(0.f, v);    // row_ror:4
    v += dpp_move<0x128, 0xF>(0.f, v);    // row_ror:8  -> every lane of a 16-lane row holds the row total
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);      // {row0,row0,row2,row2} + {row1,row1,row3,row3}
}

__device__ __forceinline__ f32x4 ld4(const float __attribute__((address_space(1))) *p, bool aligned) {
    if (aligned) return *reinterpret_cast<const f32x4 __attribute__((address_space(1))) *>(p);
    return f32x4{p[0], p[1], p[2], p[3]};
}

struct FwRow {           // what a lane needs to know about tile row m
    int b, t, nx;        // sample, quantile index, 1 = next-state row (kind 2)
    int64_t save;        // row index in the per-row save arrays, -1: not saved
};

// WAVES = 4 (bf16 path, tile kinds 0 and 1): the same tile on a 256-thread workgroup, each wave streaming 256 columns.  Two such
// workgroups fit one CU (51 KB of LDS, two waves per SIMD between them), so where a launch has several tiles per CU -- the
// full model: 1152 tiles -- the prologue, fold and row phase of one tile run beside the weight stream of another instead of
// in front of it (one 512-thread workgroup per CU leaves the matrix pipe and the L2 path idle for a third of every tile).
template <int H, bool LN, bool SPLIT = false, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES, 2) void fwd_tile_kernel(IqnArgs a_by_value) {      // (two waves per SIMD: 256 registers)
    kernarg_prefetch<sizeof(IqnArgs)>();
    static_assert(WAVES == 8 || (WAVES == 4 && SPLIT && H == 128), "four-wave tiles: bf16 path at width 128");
    constexpr int NHT = H / 16, HP = H + 4, KPT = H / 128, FW_WAVES = WAVES, NTHREADS = 64 * WAVES;
    constexpr int ROW_PASSES = 512 / NTHREADS;       // the row phase handles 16 rows x 32 lanes: two passes of eight rows on four waves
    constexpr int SPW = UV_SLICES / FW_WAVES;        // 128-column u slices per streaming wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef FwLds<H, WAVES> LD;
    float *cost = smem + LD::COST.off;           // [16][CS] cos basis of the tile's rows
    float *rowf = smem + LD::ROWF.off;           // [0,16) tau | [16,32) y | [32,48) q | [48,64) dq | [64,80) tau (loss order)
    float *stat = smem + LD::STAT.off;           // [3][8 waves][16 rows] shifted row sums / sums of squares / the shift
    float *part = smem + LD::PART.off;           // [8 waves][16 rows][HP] K-slice partials
    float *zt = smem + LD::ZT.off;               // [16][FW_ZS] quantile / Q estimates of the tile
    float *w2s = smem + LD::W2S.off;             // [A][H] head Linear weight; b2 in rowf[96, 112)

    // The pass this tile belongs to.  A runtime index into the by-value kernel argument would make the compiler
    // copy the whole argument into scratch (and turn every pointer in it into a flat pointer), so the pass
    // descriptor is read from the kernel-argument segment itself (scalar loads from constant memory) and its
    // pointers are declared global.
    typedef const float __attribute__((address_space(1))) *gcf;
    typedef float __attribute__((address_space(1))) *gf;
    (void)a_by_value;
    const auto *kargs = (const __attribute__((address_space(4))) IqnArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const __attribute__((address_space(4))) IqnArgs &a = *kargs;      // every use below reads the argument segment
    // which pass this tile belongs to: all tile counts are requested together and compared without branches (written as
    // a chain of short-circuit tests it was five scalar loads each waited for and branched on in turn)
    int tile = blockIdx.x, pi = 0;
    const int n_pass = a.n_pass;
    int nt[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) nt[i] = a.pass[i].n_tiles;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const bool adv = (pi == i) & (i + 1 < n_pass) & (tile >= nt[i]);
        tile -= adv ? nt[i] : 0;
        pi += adv ? 1 : 0;
    }
    const __attribute__((address_space(4))) IqnPass *pp = &kargs->pass[pi];
    const gcf ps_params = (gcf)pp->params, ps_wpk = (gcf)pp->wpk, ps_uv = (gcf)pp->uv;
    const gcf ps_e = (gcf)pp->e, ps_e2 = (gcf)pp->e2, ps_tau_in = (gcf)pp->tau_in, ps_tau_in2 = (gcf)pp->tau_in2;
    const gf ps_z_out = (gf)pp->z_out, ps_z_out2 = (gf)pp->z_out2;
    const int ps_T = pp->T, ps_save = pp->save, ps_stream_id = pp->stream_id, ps_kind = pp->kind;
    const int B = a.B, A = a.A;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int li = lane & 15, g = lane >> 4;
    const int kind = ps_kind;
    const int T = ps_T, tsh = 31 - __clz(T);
    const gcf P = ps_params;
    PRISM_STAMP(0);

    int hd = 0, b0 = 0;
    if (kind == 1) {
        // Workgroups go to the eight XCDs round-robin by workgroup index, and an XCD's 4 MB L2 holds the streamed weights of
        // five heads at most (786 KB each as bf16 pieces): in head-major tile order every XCD streams EVERY head and, with
        // ten of them, keeps missing (188 MB of HBM-side traffic per launch for c4).  So the tiles of this pass are dealt
        // XCD-major: the workgroups of XCD x take a contiguous run of the head-major order -- one or two heads an XCD.
        const int n = pp->n_tiles, base = (int)blockIdx.x - tile;
        const int x = (int)blockIdx.x & 7, first = (x - base) & 7;            // first pass tile of this XCD
        int before = 0;                                                       // pass tiles of the XCDs below x
#pragma unroll
        for (int y = 0; y < 7; ++y) {
            const int fy = (y - base) & 7;
            before += (y < x && fy < n) ? (n - fy + 7) >> 3 : 0;
        }
        const int v = before + ((tile - first) >> 3);
        const int tiles_per_head = B / 16;
        hd = v / tiles_per_head;
        b0 = (v - hd * tiles_per_head) * 16;
    }
    auto row_of = [&](int m) __attribute__((always_inline)) {
        FwRow r;
        r.nx = 0;
        if (kind == 1) {
            r.b = b0 + m;
            r.t = 0;
            r.save = ps_save ? (int64_t)hd * B + r.b : -1;
        } else if (kind == 0) {
            // (T need not be a power of two here: the acting forward runs 200 quantile samples per observation, and its
            // last tile may run past the last row: such rows compute on the last sample's data and land in padding)
            const int row = tile * 16 + m;
            r.b = min((T & (T - 1)) == 0 ? row >> tsh : row / T, B - 1);
            r.t = row - r.b * T;
            r.save = ps_save ? (int64_t)row : -1;
        } else {
            const int s = m >> (tsh + 1), j = m & (2 * T - 1);
            r.b = tile * (16 >> (tsh + 1)) + s;
            r.nx = j >= T;
            r.t = j & (T - 1);
            r.save = r.nx ? -1 : (int64_t)r.b * T + r.t;
        }
        return r;
    };
    const FwRow myrow = row_of(li);              // the row this lane feeds into the B operands (m = li)
    // selects between two descriptor pointers stay OUT of the lambdas below: inside one, a select of two
    // captured values becomes a load from a computed address of the closure object, which then -- with every
    // array it references -- has to live in scratch
    const gcf e_base = (kind == 2 && myrow.nx) ? ps_e2 : ps_e;
    const FwRow trow = row_of(tid & 15);         // tid < 16: the row whose quantile sample this thread draws
    const gcf tau_src = (kind == 2 && trow.nx) ? ps_tau_in2 : ps_tau_in;
    const int tau_sid = (kind == 2 && trow.nx) ? 1 : ps_stream_id;

    // the tau counter of the device RNG: a cold word the quantile draw hangs on -- requested before everything else (asked
    // for where it is used, inside the prologue's tid < 16 branch, its round trip sat in front of the Philox rounds, the
    // cosines and two barriers)
    unsigned long long rng_tau = 0ull;
    if (kind != 1 && a.rng) rng_tau = ((const unsigned long long __attribute__((address_space(1))) *)a.rng)[1];
    // head Linear weight [A][H] and bias: the oldest requests of the kernel, so that waiting for them later
    // waits for nothing else (clamped indices: unconditional loads)
    constexpr int W2N = 16 * H / NTHREADS;
    float w2r[W2N], b2r;
    {
        const gcf Ph = kind == 1 ? P + a.off.head_base + (int64_t)hd * a.off.head_stride : P;
        const gcf W2 = Ph + (kind == 1 ? a.off.h_w2 : a.off.iqn_w2), B2 = Ph + (kind == 1 ? a.off.h_b2 : a.off.iqn_b2);
#pragma unroll
        for (int i = 0; i < W2N; ++i) w2r[i] = W2[min(tid + NTHREADS * i, A * H - 1)];
        b2r = B2[min(tid, A - 1)];
    }

    f32x4 accT[NHT];
#pragma unroll
    for (int i = 0; i < NHT; ++i) accT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f, cshift = 0.f;
    static_assert(UV_SLICES % FW_WAVES == 0, "whole u slices per streaming wave");

    // ---------------------------------------------------------------------------------------------
    // the streamed products.  PHI: IQN rows (phi product feeds the trunk product); !PHI: Q-head rows.
    // Every weight register is refilled for the next 16-column step right after its last use, so each
    // wave keeps one step of its stream (4 + H/16 KB) in flight at all times.
    // ---------------------------------------------------------------------------------------------
    auto stream = [&](auto phi_tag, auto nt_tag, const int step0) __attribute__((always_inline)) {
        constexpr bool PHI = decltype(phi_tag)::value;
        constexpr int FW_NT = decltype(nt_tag)::value;      // 16-column steps of this wave, the first one being step0
        constexpr int SL = PHI ? NHT + 4 : NHT, W0 = PHI ? 4 : 0;
        typedef const f32x4 __attribute__((address_space(1))) *gcf4;      // (native vectors: a HIP float4 loaded through an
                                                                          // address-space pointer is copied via memory)
        const gcf4 wp = reinterpret_cast<gcf4>(ps_wpk) + (kind == 1 ? (size_t)hd * (H * E_DIM / 4) : (size_t)0) +
                           (size_t)step0 * SL * 64 + lane;
        const gcf erow = e_base + (int64_t)myrow.b * E_DIM + 16 * step0 + 4 * g;
        const gcf brow = P + a.off.phi_b + 16 * step0 + 4 * g;
        // first requests: everything the first column step needs, plus what the second one needs before
        // a refill of the first one's registers could land
        // trunk weights: WD register sets, each refilled for column step nt + WD right after step nt used it
        constexpr int WD = 1;      // (two steps in flight measured no faster: the loop is not load-latency-bound)
        f32x4 wphi0[4], wphi[4], w1[WD][NHT], e4[2], b4[2];
        if (PHI) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wphi0[q] = wp[q * 64];
            b4[0] = *reinterpret_cast<gcf4>(brow);
            b4[1] = *reinterpret_cast<gcf4>(brow + 16);
        }
        e4[0] = *reinterpret_cast<gcf4>(erow);
        e4[1] = *reinterpret_cast<gcf4>(erow + 16);
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) w1[0][ht] = wp[(W0 + ht) * 64];
        if (PHI) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wphi[q] = wp[(SL + q) * 64];
        }
        if (WD == 2) {
#pragma unroll
            for (int ht = 0; ht < NHT; ++ht) w1[WD - 1][ht] = wp[(SL + W0 + ht) * 64];
        }
        // the head Linear of the row phase (requested first of all): parked in LDS, read after the fold
#pragma unroll
        for (int i = 0; i < W2N; ++i)
            if (tid + NTHREADS * i < A * H) w2s[tid + NTHREADS * i] = w2r[i];
        if (tid < A) rowf[96 + tid] = b2r;

        f32x4 cosB[4];
        if (PHI) {
            // quantile samples of the 16 rows, then their cos basis (iqn_model.py:89-93) through LDS
            if (tid < 16) {
                const FwRow r = trow;
                const int sid = tau_sid;
                const gcf tin = tau_src;
                float tau;
                if (tin) {
                    tau = tin[min((int64_t)r.t, (int64_t)T - 1) * a.Bt + r.b];
                } else {
                    uint32_t rr[4];
                    Philox ph(a.seed);
                    ph(a.offset + rng_tau + (uint64_t)((int64_t)r.t * a.Bt + r.b), 0x54415530ull + (uint64_t)sid, rr);
                    tau = u32_to_unit_float(rr[0]);
                }
                if (a.tau_out && r.t < T) a.tau_out[(int64_t)sid * a.maxT * a.Bt + (int64_t)r.t * a.Bt + r.b] = tau;
                rowf[tid] = tau;
            }
            lds_barrier();
            // c[m][k] = cos(tau * (k+1) * pi), two fp32 multiplies as torch does (iqn_model.py:90-92)
            for (int idx = tid; idx < 16 * K_BASIS; idx += NTHREADS) {
                const int m = idx >> 6, k = idx & 63;
                const float c = cosf((rowf[m] * (float)(k + 1)) * PI_F);
                cost[m * CS + k] = c;
                const FwRow r = row_of(m);
                if (r.save >= 0) __builtin_nontemporal_store(c, &a.ws.cosb[r.save * K_BASIS + k]);   // the backward launch reads it
            }
            lds_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) cosB[q] = *reinterpret_cast<const f32x4 *>(&cost[li * CS + 16 * q + 4 * g]);
        }
        PRISM_STAMP(1);

        float x[4], xn[4];
        f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#define FW_SEL(v, c) ((v)[c])
        // phi epilogue of column step `nt_e` (or the head rows' input); rolls the e / bias registers forward
        // component r of column step nt_e's input; after the last component the e registers roll forward
        // ReLU(phi) of the rows the backward will see is saved as it is produced: 16 bytes per lane and step (the lane's
        // four columns of its row), blocked [row / 16][column step][row][column] so that a backward workgroup finds the
        // 16 x 16 block of its columns and tile in one kilobyte.  It saves that launch a sixth of its matrix work (the phi
        // product it used to recompute) for 8 MB written here and read there.
        // (Kept in registers until the stream is over -- 32 of them, which two waves per SIMD have to spare: memory
        // operations of a wave retire in order, a store in the middle of the stream makes every later wait for a weight
        // refill wait for the store's acknowledgement as well: +1.5 us, measured.)
        // Width 256 has no registers left (237 -- even the pointer spills): nothing is saved there, the backward recomputes.
        constexpr bool PH_ALL = H == 128;
        f32x4 phs[PH_ALL ? FW_NT : 1];
        float *const phi_dst = (PH_ALL && PHI && myrow.save >= 0)
                                   ? a.ws.phis + ((myrow.save >> 4) * (int64_t)(E_DIM / 16) + step0) * 256 + (myrow.save & 15) * 16 + 4 * g
                                   : nullptr;
        auto finish_x = [&](float (&dst)[4], int nt_e, int r) __attribute__((always_inline)) {
            const float ev = e4[nt_e & 1][r];
            const float relu = PHI ? fmaxf(pacc[r] + b4[nt_e & 1][r], 0.f) : 0.f;
            if (PHI && PH_ALL) phs[nt_e][r] = relu;
            float xv = PHI ? relu * ev : ev;
            if (LN) {
                // the wave's shift of row li: the first element its g = 0 lane produces (all four lanes of the row agree)
                if (nt_e == 0 && r == 0) cshift = __shfl(xv, li, 64);
                xv -= cshift;
                s1 += xv;
                s2 = fmaf(xv, xv, s2);
            }
            dst[r] = xv;
            if (r == 3 && nt_e + 2 < FW_NT) {
                e4[nt_e & 1] = *reinterpret_cast<gcf4>(erow + 16 * (nt_e + 2));
                if (PHI) b4[nt_e & 1] = *reinterpret_cast<gcf4>(brow + 16 * (nt_e + 2));
            }
        };
        if (PHI) {
#pragma unroll
            for (int i = 0; i < 16; ++i) pacc = mfma16(FW_SEL(wphi0[i >> 2], i & 3), FW_SEL(cosB[i >> 2], i & 3), pacc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) finish_x(x, 0, r);
        // The instruction order below IS the schedule: every slot is fenced, because left alone the scheduler
        // sinks each weight load to just before its first use (trading the whole prefetch for registers
        // nobody needs) and strings the links of the phi chain together (40-cycle stalls).
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < FW_NT; ++nt) {
            const bool more = nt + 1 < FW_NT;

            if (PHI && more) pacc = f32x4{0.f, 0.f, 0.f, 0.f};
            // trunk MFMAs of this step; the phi chain of the NEXT step is threaded between them (one phi MFMA
            // behind each trunk MFMA until it is done: consecutive links of the chain are then two issue
            // slots apart, more than the 40-cycle accumulator latency)
#pragma unroll
            for (int hp = 0; hp < NHT / 2; ++hp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // phi steps carried by this (hp, r) slot: two while H = 128 (16 slots carry 16 + 16), one for H = 256
                    const int p0 = NHT == 8 ? (hp * 4 + r) * 2 : hp * 4 + r, p1 = NHT == 8 ? p0 + 1 : 16;
                    accT[2 * hp] = mfma16(FW_SEL(w1[nt % WD][2 * hp], r), x[r], accT[2 * hp]);
                    if (PHI && more && p0 < 16) {
                        pacc = mfma16(FW_SEL(wphi[p0 >> 2], p0 & 3), FW_SEL(cosB[p0 >> 2], p0 & 3), pacc);
                        if ((p0 & 3) == 3 && nt + 2 < FW_NT) wphi[p0 >> 2] = wp[((nt + 2) * SL + (p0 >> 2)) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    accT[2 * hp + 1] = mfma16(FW_SEL(w1[nt % WD][2 * hp + 1], r), x[r], accT[2 * hp + 1]);
                    if (PHI && more && p1 < 16) {
                        pacc = mfma16(FW_SEL(wphi[p1 >> 2], p1 & 3), FW_SEL(cosB[p1 >> 2], p1 & 3), pacc);
                        if ((p1 & 3) == 3 && nt + 2 < FW_NT) wphi[p1 >> 2] = wp[((nt + 2) * SL + (p1 >> 2)) * 64];
                    }
                    // the next step's input, one component per slot: a pair of trunk tiles after the phi chain
                    // ended (its last MFMA has drained by then); head rows: in the last slots, when their e
                    // registers have had a whole step to land
                    if (more && hp == (PHI ? (NHT == 8 ? 2 : 4) : NHT / 2 - 1)) finish_x(xn, nt + 1, r);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (nt + WD < FW_NT) {
                    w1[nt % WD][2 * hp] = wp[((nt + WD) * SL + W0 + 2 * hp) * 64];
                    w1[nt % WD][2 * hp + 1] = wp[((nt + WD) * SL + W0 + 2 * hp + 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (more) {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = xn[r];
            }
        }
        if (PH_ALL && PHI && phi_dst) {
#pragma unroll
            for (int nt = 0; nt < FW_NT; ++nt) __builtin_nontemporal_store(phs[nt], reinterpret_cast<f32x4 *>(phi_dst + 256 * nt));
        }
#undef FW_SEL
    };

    // ---------------------------------------------------------------------------------------------
    // SPLIT: the same products on the bf16 matrix pipe (common.h: three-piece operands, six piece products, fp32
    // accumulation -- fp32 accuracy at 2.5x the matrix rate, and the vector unit is free half of every MFMA's cycles).
    // A wave walks its 128 columns as four 32-column double steps (K = 32 per MFMA).  Weights arrive pre-split from the
    // packing role (iqn_kernels.h pack_split_block): 36 KB per double step for an IQN tile, through two small register
    // rings (trunk: RW operand groups in flight, phi: 2), each slot refilled right after its last use; the cos basis is
    // split once per tile, the trunk input (8 values per lane and double step) as the phi epilogue produces it.  The K
    // index of the trunk operands is permuted so that lane group g's eight elements ARE its accumulator registers of the
    // two phi n-tiles.  Dependent bf16 MFMAs on one accumulator issue back to back, so no chain threading is needed; the
    // phi groups of double step d + 1 sit between the trunk groups of d only to spread the loads.
    // ---------------------------------------------------------------------------------------------
    auto stream_split = [&](auto phi_tag) __attribute__((always_inline)) {
        constexpr bool PHI = decltype(phi_tag)::value;
        constexpr int G = (PHI ? 4 : 0) + NHT, P0 = PHI ? 4 : 0, NDS = 32 / FW_WAVES, RW = FW_SPLIT_RING, NQ = NDS * NHT, NP = NDS * 4;
        constexpr int WCOLS = E_DIM / FW_WAVES;          // embed columns of a wave
        typedef const u32x4 __attribute__((address_space(1))) *gcu4;
        typedef const f32x4 __attribute__((address_space(1))) *gcf4;
        const gcu4 wp = reinterpret_cast<gcu4>(ps_wpk) + (kind == 1 ? (size_t)hd * (32 * NHT * 3 * 64) : (size_t)0) +
                        (size_t)(NDS * w) * G * 3 * 64 + lane;
        auto slot = [&](int ds, int group, int plane) __attribute__((always_inline)) { return wp[((ds * G + group) * 3 + plane) * 64]; };
        constexpr bool KO_HALF = (FW_KO & 4) != 0;      // (wrong numbers: the second half of the stream re-uses stale operand registers)
        const gcf erow = e_base + (int64_t)myrow.b * E_DIM + WCOLS * w + 4 * g;
        const gcf brow = P + a.off.phi_b + WCOLS * w + 4 * g;
        // prepared quantile samples + basis pieces of this tile (cos_basis_block): the first requests of the stream
        const unsigned int __attribute__((address_space(1))) *cpk =
            PHI ? (const unsigned int __attribute__((address_space(1))) *)pp->cospk : nullptr;
        unsigned int c_h[ROW_PASSES], c_m[ROW_PASSES], c_l[ROW_PASSES], tbits = 0;
#pragma unroll
        for (int u = 0; u < ROW_PASSES; ++u) c_h[u] = c_m[u] = c_l[u] = 0;
        if (PHI && cpk) {
            const unsigned int __attribute__((address_space(1))) *blk = cpk + (size_t)tile * CP_TILE + tid;
#pragma unroll
            for (int u = 0; u < ROW_PASSES; ++u) {
                c_h[u] = blk[NTHREADS * u];
                c_m[u] = blk[16 * 32 + NTHREADS * u];
                c_l[u] = blk[2 * 16 * 32 + NTHREADS * u];
            }
            if (tid < 16) tbits = cpk[(size_t)tile * CP_TILE + 3 * 16 * 32 + tid];
        }
        u32x4 wph[2][3], w1r[RW][3];
        f32x4 e4[2], b4[2];                    // [n-tile] of the double step whose epilogue comes next (refilled by it: one
                                               // double step, 2 k cycles, ahead of their use)
        if (PHI) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wph[p][pl] = slot(0, p, pl);
#pragma unroll
            for (int t = 0; t < 2; ++t) b4[t] = *reinterpret_cast<gcf4>(brow + 16 * t);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) e4[t] = *reinterpret_cast<gcf4>(erow + 16 * t);
#pragma unroll
        for (int q = 0; q < RW; ++q)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w1r[q][pl] = slot(0, P0 + q, pl);
        // the head Linear of the row phase (requested first of all): parked in LDS, read after the fold
#pragma unroll
        for (int i = 0; i < W2N; ++i)
            if (tid + NTHREADS * i < A * H) w2s[tid + NTHREADS * i] = w2r[i];
        if (tid < A) rowf[96 + tid] = b2r;

        unsigned int *cosp = reinterpret_cast<unsigned int *>(cost);       // [3][16][FW_CBS] dwords of bf16 pairs
        if (PHI && cpk) {
            // quantile samples and basis pieces prepared by the embed / front launch (cos_basis_block): one round of loads
            // and one barrier where the tile's own draw + cosines + split were 7 k cycles in front of its first MFMA
#pragma unroll
            for (int u = 0; u < ROW_PASSES; ++u) {
                const int idx = tid + NTHREADS * u, m = idx >> 5, kp = idx & 31;          // one basis pair per (thread, pass)
                cosp[(0 * 16 + m) * FW_CBS + kp] = c_h[u];
                cosp[(1 * 16 + m) * FW_CBS + kp] = c_m[u];
                cosp[(2 * 16 + m) * FW_CBS + kp] = c_l[u];
            }
            if (tid < 16) rowf[tid] = __uint_as_float(tbits);
            lds_barrier();
        } else if (PHI) {
            if (tid < 16) {
                const FwRow r = trow;
                const int sid = tau_sid;
                const gcf tin = tau_src;
                float tau;
                if (tin) {
                    tau = tin[min((int64_t)r.t, (int64_t)T - 1) * a.Bt + r.b];
                } else {
                    uint32_t rr[4];
                    Philox ph(a.seed);
                    ph(a.offset + rng_tau + (uint64_t)((int64_t)r.t * a.Bt + r.b), 0x54415530ull + (uint64_t)sid, rr);
                    tau = u32_to_unit_float(rr[0]);
                }
                if (a.tau_out && r.t < T) a.tau_out[(int64_t)sid * a.maxT * a.Bt + (int64_t)r.t * a.Bt + r.b] = tau;
                rowf[tid] = tau;
            }
            lds_barrier();
#pragma unroll
            for (int u = 0; u < ROW_PASSES; ++u) {
                // (thread, pass) = (row m, basis pair k0, k0 + 1): the cos values as torch computes them (iqn_model.py:90-92), saved
                // in fp32 for the backward launch, and their three bf16 pieces parked as the B operand of the phi product
                const int idx = tid + NTHREADS * u, m = idx >> 5, k0 = 2 * (idx & 31);
                const float tm = rowf[m];
                const float c0 = cosf((tm * (float)(k0 + 1)) * PI_F), c1 = cosf((tm * (float)(k0 + 2)) * PI_F);
                const FwRow r = row_of(m);
                if (r.save >= 0) {
                    __builtin_nontemporal_store(c0, &a.ws.cosb[r.save * K_BASIS + k0]);
                    __builtin_nontemporal_store(c1, &a.ws.cosb[r.save * K_BASIS + k0 + 1]);
                }
                const unsigned int h = pack_bf16(c0, c1);
                const float ra = c0 - __uint_as_float(h << 16), rb = c1 - __uint_as_float(h & 0xffff0000u);
                const unsigned int md = pack_bf16(ra, rb);
                const float sa = ra - __uint_as_float(md << 16), sb = rb - __uint_as_float(md & 0xffff0000u);
                cosp[(0 * 16 + m) * FW_CBS + (idx & 31)] = h;
                cosp[(1 * 16 + m) * FW_CBS + (idx & 31)] = md;
                cosp[(2 * 16 + m) * FW_CBS + (idx & 31)] = pack_bf16(sa, sb);
            }
            lds_barrier();
        }
        // B operand of the phi product, K block kb: cos[m = li][k = 32 kb + 8 g + j] -- read from LDS at every use (three
        // 16-byte reads per six MFMAs: free beside them, and 24 registers go to the weight ring instead)
        auto cos_operand = [&](int kb) __attribute__((always_inline)) {
            Split3 c;
            c.hi = *reinterpret_cast<const u32x4 *>(&cosp[(0 * 16 + li) * FW_CBS + 16 * kb + 4 * g]);
            c.mid = *reinterpret_cast<const u32x4 *>(&cosp[(1 * 16 + li) * FW_CBS + 16 * kb + 4 * g]);
            c.lo = *reinterpret_cast<const u32x4 *>(&cosp[(2 * 16 + li) * FW_CBS + 16 * kb + 4 * g]);
            return c;
        };
        PRISM_STAMP(1);

        f32x4 pacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        Split3 xs;
        float *const phi_dst = (PHI && myrow.save >= 0)
                                   ? a.ws.phis + ((myrow.save >> 4) * (int64_t)(E_DIM / 16) + (WCOLS / 16) * w) * 256 + (myrow.save & 15) * 16 + 4 * g
                                   : nullptr;
        // (ReLU(phi) for the backward launch goes out from inside the stream, 16 bytes per lane and n-tile.  Knock-out, same
        // box: without these stores the tile takes 20.9 instead of 22.4 us -- but parking the values in LDS and storing them
        // behind the last MFMA group only moved the cost: forward -0.3 us, backward +0.2 us (r04): what costs is the 8 MB
        // that have to be out before the launch ends, not the stores' place in the wave's memory queue.)
        // phi group p (0 .. 4 NDS - 1): double step p >> 2, n-tile (p >> 1) & 1, K block p & 1; ring slot p & 1
        auto phi_group = [&](int p) __attribute__((always_inline)) {
            const int nt2 = (p >> 1) & 1, kb = p & 1;
            pacc[nt2] = mfma_split(wph[p & 1][0], wph[p & 1][1], wph[p & 1][2], cos_operand(kb), pacc[nt2]);
            if (p + 2 < (KO_HALF ? NP / 2 : NP)) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wph[p & 1][pl] = slot((p + 2) >> 2, (p + 2) & 3, pl);
            }
        };
        // the trunk input of double step ds from its two phi accumulators (or, head rows, from e), shifted, split
        auto epilogue = [&](int ds) __attribute__((always_inline)) {
            float xv[8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 ph4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ev = e4[t][r];
                    const float relu = PHI ? fmaxf(pacc[t][r] + b4[t][r], 0.f) : 0.f;
                    ph4[r] = relu;
                    float v = PHI ? relu * ev : ev;
                    if (LN) {
                        if (ds == 0 && t == 0 && r == 0) cshift = __shfl(v, li, 64);
                        v -= cshift;
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                    xv[4 * t + r] = v;
                }
                if (PHI && phi_dst && !(FW_KO & 1)) __builtin_nontemporal_store(ph4, reinterpret_cast<f32x4 *>(phi_dst + 256 * (2 * ds + t)));
                if (ds + 1 < NDS) {
                    e4[t] = *reinterpret_cast<gcf4>(erow + 32 * (ds + 1) + 16 * t);
                    if (PHI) b4[t] = *reinterpret_cast<gcf4>(brow + 32 * (ds + 1) + 16 * t);
                }
                if (PHI) pacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr ((FW_KO & 2) != 0) {
                if (ds == 0) xs = split_bf16x3(xv);
                else asm volatile("" ::"v"(xv[0]), "v"(xv[1]), "v"(xv[2]), "v"(xv[3]), "v"(xv[4]), "v"(xv[5]), "v"(xv[6]), "v"(xv[7]));
            } else
                xs = split_bf16x3(xv);
        };
        if (PHI) {
#pragma unroll
            for (int p = 0; p < 4; ++p) phi_group(p);
        }
        epilogue(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
#pragma unroll
            for (int ht = 0; ht < NHT; ++ht) {
                const int q = ds * NHT + ht, sl = q % RW;
                accT[ht] = mfma_split(w1r[sl][0], w1r[sl][1], w1r[sl][2], xs, accT[ht]);
                if (q + RW < (KO_HALF ? NQ / 2 : NQ)) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) w1r[sl][pl] = slot((q + RW) / NHT, P0 + (q + RW) % NHT, pl);
                }
                __builtin_amdgcn_sched_barrier(0);
                // a phi group of the next double step behind every second trunk group (NHT = 8: four of them)
                if (PHI && ds + 1 < NDS && (ht & 1) && (ht >> 1) < 4) {
                    phi_group(4 * (ds + 1) + (ht >> 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (ds + 1 < NDS) {
                epilogue(ds + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if constexpr (SPLIT) {
        static_assert((H == 128 && (FW_WAVES == 8 || FW_WAVES == 4)) || (H == 256 && FW_WAVES == 8),
                      "the bf16 path: width 128 on eight or four streaming waves, width 256 on eight");
        if (kind == 1) stream_split(std::false_type{});
        else stream_split(std::true_type{});
    } else
    {
        auto run = [&](auto phi_tag) __attribute__((always_inline)) {
            if constexpr (FW_WAVES == 12) {
                if (w < 4) stream(phi_tag, std::integral_constant<int, 6>{}, 6 * w);
                else stream(phi_tag, std::integral_constant<int, 5>{}, 24 + 5 * (w - 4));
            } else {
                stream(phi_tag, std::integral_constant<int, FW_STEPS / FW_WAVES>{}, (FW_STEPS / FW_WAVES) * w);
            }
        };
        if (kind == 1) run(std::false_type{});
        else run(std::true_type{});
    }
    PRISM_STAMP(2);

    // ---- fold the K-slices ------------------------------------------------------------------------
    const int fc = tid & 31;                         // fold / row phase: hidden units 4 fc .. 4 fc + 3 (+128) of row fm
    // operands of the row phase: requested now, consumed after the barrier
    const gcf Ptr = kind == 1 ? P + a.off.head_base + (int64_t)hd * a.off.head_stride : P;
    const int64_t o_b1 = kind == 1 ? a.off.h_b1 : a.off.iqn_b1, o_g2 = kind == 1 ? a.off.h_ln2_g : a.off.iqn_ln2_g;
    const int64_t o_be2 = kind == 1 ? a.off.h_ln2_b : a.off.iqn_ln2_b;
    const bool al = kind != 1;                       // head tensors are only 4-byte aligned
    const gcf uvp = ps_uv + (kind == 1 ? (size_t)hd * UV_ROWS * H : (size_t)0);
    f32x4 u4[KPT], vb4[KPT], g24[KPT], be24[KPT], us4[LN ? FW_WAVES : 1][KPT];      // us4[w]: u over the columns wave w streamed
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int h0 = 128 * k + 4 * fc;
        const f32x4 bb = ld4(Ptr + o_b1 + h0, al);
        if (LN) {
            u4[k] = ld4(uvp + h0, true);
            vb4[k] = ld4(uvp + H + h0, true) + bb;
#pragma unroll
            for (int ww = 0; ww < FW_WAVES; ++ww) {
                f32x4 t = ld4(uvp + (2 + SPW * ww) * H + h0, true);
#pragma unroll
                for (int sl = 1; sl < SPW; ++sl) t += ld4(uvp + (2 + SPW * ww + sl) * H + h0, true);
                us4[LN ? ww : 0][k] = t;
            }
            g24[k] = ld4(Ptr + o_g2 + h0, al);
            be24[k] = ld4(Ptr + o_be2 + h0, al);
        } else {
            vb4[k] = bb;
        }
    }
    if (LN) {
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (g == 0) {
            stat[w * 16 + li] = s1;
            stat[(FW_WAVES + w) * 16 + li] = s2;
            stat[(2 * FW_WAVES + w) * 16 + li] = cshift;
        }
    }
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht)      // accT[ht][r] = pre^T[h = 16 ht + 4 g + r][m = li]
        *reinterpret_cast<float4 *>(&part[(w * 16 + li) * HP + 16 * ht + 4 * g]) = float4{accT[ht][0], accT[ht][1], accT[ht][2], accT[ht][3]};
    if (FW_WAVES > FW_ROW_WAVES && w >= FW_ROW_WAVES) {      // streaming-only waves: partial sums are out, done
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }
    lds_barrier();
    PRISM_STAMP(3);

    // The row phase: one row per 32-lane half-wave -- all sixteen at once on eight waves, two passes of eight on four.
    // (What the loss tail of a mixed tile reads of it -- pre, hx, rstd2, frow, fm of EVERY row at once -- only exists with
    // ROW_PASSES == 1: mixed tiles run on eight waves.)
    float pre[4 * KPT], hx[4 * KPT];           // pre-activation / what feeds the head Linear (xhat2 with LN, ReLU(pre) without)
    float rstd2 = 1.f;
    int fm = tid >> 5;
    FwRow frow = row_of(fm);
    const bool local_loss = kind == 2;
#pragma unroll
    for (int rp = 0; rp < ROW_PASSES; ++rp) {
    if (rp > 0) {
        fm = (tid >> 5) + (NTHREADS / 32) * rp;
        frow = row_of(fm);
    }
    float mean1 = 0.f, rstd1 = 1.f;
    float cdev[LN ? FW_WAVES : 1];             // c_w - mean of this row: what multiplies the slice sums u_w below
    if (LN) {
        // shifted moments of the K slices (NW elements each): mean_w = c_w + s1_w / NW, M2_w = s2_w - s1_w^2 / NW;
        // whole row: mean = avg(mean_w), M2 = sum M2_w + NW sum (mean_w - mean)^2
        constexpr float NW = (float)(E_DIM / FW_WAVES);
        float cw[FW_WAVES], a1[FW_WAVES], a2[FW_WAVES], cbar = 0.f, t1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < FW_WAVES; ++ww) {
            a1[ww] = stat[ww * 16 + fm];
            a2[ww] = stat[(FW_WAVES + ww) * 16 + fm];
            cw[ww] = stat[(2 * FW_WAVES + ww) * 16 + fm];
        }
#pragma unroll
        for (int ww = 0; ww < FW_WAVES; ++ww) {
            cbar += cw[ww];
            t1 += a1[ww];
        }
        cbar *= 1.0f / FW_WAVES;
        const float dmean = t1 * (1.0f / E_DIM);          // mean - cbar
        mean1 = cbar + dmean;
        float m2 = 0.f;
#pragma unroll
        for (int ww = 0; ww < FW_WAVES; ++ww) {
            const float dw = (cw[ww] - cbar) + (a1[ww] * (1.0f / NW) - dmean);      // mean_w - mean
            m2 += (a2[ww] - a1[ww] * a1[ww] * (1.0f / NW)) + NW * (dw * dw);
            cdev[LN ? ww : 0] = (cw[ww] - cbar) - dmean;
        }
        const float var = fmaxf(m2 * (1.0f / E_DIM), 0.f);
        rstd1 = 1.0f / sqrtf(var + LN_EPS);
        if (frow.save >= 0 && fc == 0) {
            (kind == 1 ? a.ws.q_mu1 : a.ws.mu1)[frow.save] = mean1;
            (kind == 1 ? a.ws.q_rstd1 : a.ws.rstd1)[frow.save] = rstd1;
        }
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int h0 = 128 * k + 4 * fc;
        float4 s = *reinterpret_cast<const float4 *>(&part[fm * HP + h0]);
#pragma unroll
        for (int ww = 1; ww < FW_WAVES; ++ww) {
            const float4 p = *reinterpret_cast<const float4 *>(&part[(ww * 16 + fm) * HP + h0]);
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        float sv[4] = {s.x, s.y, s.z, s.w};
        if (LN) {
            // the shifts come back: sum_w (c_w - mean) u_w[h]   (slices in order)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = 0.f;
#pragma unroll
                for (int ww = 0; ww < FW_WAVES; ++ww) t = fmaf(cdev[LN ? ww : 0], us4[LN ? ww : 0][k][c], t);
                sv[c] += t;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) pre[4 * k + c] = LN ? rstd1 * sv[c] + vb4[k][c] : sv[c] + vb4[k][c];
    }
    // ---- ReLU -> [LayerNorm(H)] -> head --------------------------------------------------------------
    rstd2 = 1.f;
    {
        float hs = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * KPT; ++i) {
            hx[i] = fmaxf(pre[i], 0.f);
            hs += hx[i];
        }
        if (LN) {
            const float mean2 = half_sum(hs) * (1.0f / H);
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < 4 * KPT; ++i) {
                hx[i] -= mean2;
                vs = fmaf(hx[i], hx[i], vs);
            }
            rstd2 = 1.0f / sqrtf(half_sum(vs) * (1.0f / H) + LN_EPS);
#pragma unroll
            for (int i = 0; i < 4 * KPT; ++i) hx[i] *= rstd2;
        }
    }
    if (frow.save >= 0 && !local_loss) {
        // saved for the loss / backward launches: pre-activation and the head Linear's input
        float *sp = (kind == 1 ? a.ws.q_pre1 : a.ws.pre1) + frow.save * H, *sx = (kind == 1 ? a.ws.q_xhat2 : a.ws.xhat2) + frow.save * H;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int h0 = 128 * k + 4 * fc;
            stream_store4(reinterpret_cast<float4 *>(sp + h0), float4{pre[4 * k], pre[4 * k + 1], pre[4 * k + 2], pre[4 * k + 3]});
            stream_store4(reinterpret_cast<float4 *>(sx + h0), float4{hx[4 * k], hx[4 * k + 1], hx[4 * k + 2], hx[4 * k + 3]});
        }
        if (LN && fc == 0) (kind == 1 ? a.ws.q_rstd2 : a.ws.rstd2)[frow.save] = rstd2;
    }
    {
        float y[4 * KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
#pragma unroll
            for (int c = 0; c < 4; ++c) y[4 * k + c] = LN ? hx[4 * k + c] * g24[k][c] + be24[k][c] : hx[4 * k + c];
        }
        float zmine = 0.f;
#pragma unroll
        for (int aa = 0; aa < 16; ++aa)
            if (aa < A) {
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < KPT; ++k) {
                    const float4 wv = *reinterpret_cast<const float4 *>(&w2s[aa * H + 128 * k + 4 * fc]);
                    d += (y[4 * k] * wv.x + y[4 * k + 1] * wv.y) + (y[4 * k + 2] * wv.z + y[4 * k + 3] * wv.w);
                }
                const float z = half_sum(d);
                if (fc == aa) zmine = z;
            }
        if (fc < A) {
            zmine += rowf[96 + fc];
            zt[fm * FW_ZS + fc] = zmine;
            // quantile / Q estimates of the row (also what the stand-alone loss kernels read)
            const gf zo = (kind == 2 && frow.nx) ? ps_z_out2 : ps_z_out;
            const int64_t zr = kind == 1 ? (int64_t)hd * B + frow.b : (int64_t)frow.b * T + frow.t;      // (kind 0: == row)
            zo[zr * A + fc] = zmine;
        }
    }
    }      // row passes
    PRISM_STAMP(4);
    if (!local_loss) return;
    if constexpr (ROW_PASSES != 1) return;           // (mixed tiles are launched on eight waves only)

    // =================================================================================================
    // kind 2: the loss of the tile's samples (iqn_model.py:95-201), then head + LayerNorm backward of their
    // current-state rows, straight from the registers that still hold pre / xhat2 of every row.
    // =================================================================================================
    float *s_y = rowf + 16, *s_q = rowf + 32, *s_dq = rowf + 48;
    const int ns = 16 >> (tsh + 1);                   // samples in this tile
    // operands that do not depend on the loss: requested before the barrier
    const int act_f = (int)a.action[frow.b];
    float w2a[4 * KPT], uu[4 * KPT], vbv[4 * KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const float4 wv = *reinterpret_cast<const float4 *>(&w2s[act_f * H + 128 * k + 4 * fc]);
        const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            w2a[4 * k + c] = LN ? wq[c] * g24[k][c] : wq[c];
            uu[4 * k + c] = LN ? u4[k][c] : 0.f;
            vbv[4 * k + c] = vb4[k][c];
        }
    }
    // (the sample's scalars too: behind the barrier their round trip -- cold words the front launch streamed out -- was
    // half of the loss phase)
    const int b_l = tile * ns + (w < ns ? w : 0);
    const int act_l = (int)a.action[b_l];
    const float R_l = a.reward[b_l], gam_l = a.gamma[b_l];
    const bool nt_l = a.nonterminal[b_l] != 0;
    const float wb_l = a.per_weights ? a.per_weights[b_l] : 1.0f;
    lds_barrier();
    if (w < ns) {
        // one wave per sample: a* = argmax_a mean_j Zon[j][a] (first maximum wins, iqn_model.py:129-133)
        const int smp = w, rb = smp * 2 * T, b = b_l;
        const int act = act_l;
        const float R = R_l;
        const float dg = gam_l * (nt_l ? 1.0f : 0.0f);
        const float wb = wb_l;
        const float kap = a.huber_k;
        float mean = 0.f;
        {
            // (T <= 8 here: all reads issue together, then the adds run in the sequential order)
            float zv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) zv[j] = zt[(rb + T + (j < T ? j : 0)) * FW_ZS + (lane < A ? lane : 0)];
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < T) sacc += zv[j];
            mean = sacc / (float)T;
        }
        int astar = 0;
        float bestv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean), 0));
#pragma unroll
        for (int aa = 1; aa < 16; ++aa) {
            const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean), aa));
            if (aa < A && m > bestv) {
                bestv = m;
                astar = aa;
            }
        }
        float *y = s_y + smp * T, *q = s_q + smp * T;
        if (lane < T) {
            y[lane] = td_target(a.squish, R, zt[(rb + T + lane) * FW_ZS + astar], dg);      // separate mul and add (iqn_model.py:141-148)
            q[lane] = zt[(rb + lane) * FW_ZS + act];
        }
        __builtin_amdgcn_wave_barrier();
        // pairwise quantile-Huber tile: pair p = j*T + t; a lane keeps a fixed t because T | 64
        float lsum = 0.f, gq = 0.f;
        const int t_l = lane & (T - 1);
        const float q_l = q[t_l], tau_l = rowf[rb + t_l];
        for (int p = lane; p < T * T; p += 64) {
            const int j = p >> tsh;
            const float d = y[j] - q_l;
            const float ad = fabsf(d);
            const float hub = (ad <= kap) ? 0.5f * (d * d) : kap * (ad - 0.5f * kap);
            const float wgt = fabsf(tau_l - (d < 0.f ? 1.0f : 0.0f));
            lsum += (wgt * hub) / kap;
            const float cl = fminf(fmaxf(d, -kap), kap);
            gq += (wgt * cl) / kap;
        }
        lsum = wave_sum(lsum);
        for (int o = 32; o >= T; o >>= 1) gq += __shfl_xor(gq, o, 64);
        const float dl = (lsum / (float)T) * a.dist_w;
        const float scale = -(wb / (float)B) * a.dist_w / (float)T;
        if (lane < T) s_dq[rb + lane] = gq * scale;
        if (lane == 0) {
            a.out_dl[b] = dl;
            if (a.out_td && a.n_heads == 0) a.out_td[b] = dl;  // IQN only: td_errors = distribution_loss (composite_model.py:138-139)
            a.ws.lossw[b] = dl * wb;
        }
    }
    lds_barrier();
    PRISM_STAMP(5);
    // head + LayerNorm(H) backward of row fm (current-state rows only; the other rows contribute nothing)
    float *sS = smem + LD::LOSS_S.off, *sP = smem + LD::LOSS_P.off;      // [16 rows][HP]: dq * (head input), dpre1
    {
        const bool cur = !frow.nx;
        const float dq = cur ? s_dq[fm] : 0.f;
        float da[4 * KPT], ga[4 * KPT];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * KPT; ++i) {
            da[i] = dq * w2a[i];
            m1 += da[i];
            m2 = fmaf(da[i], hx[i], m2);
        }
        if (LN) {
            m1 = half_sum(m1) * (1.0f / H);
            m2 = half_sum(m2) * (1.0f / H);
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * KPT; ++i) {
            const float gv = LN ? rstd2 * (da[i] - m1 - hx[i] * m2) : da[i];
            ga[i] = pre[i] > 0.f ? gv : 0.f;
            c1 = fmaf(ga[i], uu[i], c1);
            c2 = fmaf(ga[i], pre[i] - vbv[i], c2);
        }
        if (LN) {
            c1 = half_sum(c1);
            c2 = half_sum(c2);
        }
        if (cur) {
            float *dp = a.ws.dpre1 + frow.save * H;
#pragma unroll
            for (int k = 0; k < KPT; ++k)
                stream_store4(reinterpret_cast<float4 *>(dp + 128 * k + 4 * fc),
                              float4{ga[4 * k], ga[4 * k + 1], ga[4 * k + 2], ga[4 * k + 3]});
            if (fc == 0) {
                if (LN) {
                    a.ws.c1[frow.save] = c1;
                    a.ws.c2[frow.save] = c2;
                }
                a.ws.dq[frow.save] = dq;
            }
        }
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int h0 = 128 * k + 4 * fc;
            *reinterpret_cast<float4 *>(&sS[fm * HP + h0]) = float4{dq * hx[4 * k], dq * hx[4 * k + 1], dq * hx[4 * k + 2], dq * hx[4 * k + 3]};
            *reinterpret_cast<float4 *>(&sP[fm * HP + h0]) = float4{ga[4 * k], ga[4 * k + 1], ga[4 * k + 2], ga[4 * k + 3]};
        }
    }
    lds_barrier();
    // per sample: S_b[h] = sum_t dq_t x_t[h], P_b[h] = sum_t dpre1_t[h], D_b = sum_t dq_t (rows in order)
    for (int o = tid; o < ns * 2 * H; o += 512) {
        const int smp = o / (2 * H), k = o - smp * 2 * H, which = k >= H, h = which ? k - H : k;
        const float *src = (which ? sP : sS) + (smp * 2 * T) * HP + h;
        float tv[8], t = 0.f;
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) tv[tt] = src[(tt < T ? tt : 0) * HP];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
            if (tt < T) t += tv[tt];
        (which ? a.ws.Pb : a.ws.Sb)[(int64_t)(tile * ns + smp) * H + h] = t;
    }
    if (tid < ns) {
        float t = 0.f;
        for (int tt = 0; tt < T; ++tt) t += s_dq[tid * 2 * T + tt];
        a.ws.Db[tile * ns + tid] = t;
    }
    PRISM_STAMP(6);
}

}  // namespace prism
