// attn_mfma.h — ViT self-attention on the matrix cores (bf16 in, fp32 accumulate): ONE wave64 per (frame, head).
//
// The per-frame problem is tiny and odd (50 tokens x 64 dims per head, 16 heads: vit-pytorch Attention at
// reference model/trajectory_model.py:54-65): tokens are padded to 64 so the whole head is a 2x2 grid of 32x32 MFMA
// tiles, Q/K/V (and dO) of the head are staged once into wave-private LDS tiles (natural [token][d] layout,
// 16-byte coalesced copies, zero-filled padding rows) and everything else stays in registers:
//
//   forward   S^T = K Q^T          (lane = query column, regs = keys: the softmax row reduction is lane-local
//                                    plus one xor-32 exchange — no LDS round trip for P)
//             P   = softmax         (normalised before PV so no per-row rescale of O is needed)
//             O   = P V             (A = P straight from the S^T accumulator registers: any k-slot <-> key
//                                    assignment is legal as long as V's B fragment uses the same one, so V is read
//                                    with two ds_read_b64_tr_b16 per fragment at rows 4h+16s+{0..3, 8..11})
//   backward  both orientations are recomputed on the matrix cores (cheaper than transposing dS through LDS):
//             lane = query:  S^T, dP^T = V dO^T -> D = sum_k P dP, dS^T -> dQ = dS K
//             lane = key  :  S,   dP   = dO V^T -> P, dS (lse / D per register row from LDS) -> dV = P^T dO, dK = dS^T Q
//   32 MFMAs forward, 112 backward per head, against ~6.6 / 16.5 MFLOP of useful work (22 % padding waste).
#pragma once
#include "vc_rt.h"
#include "attn.h"

constexpr int AM_T = 64;          // padded tokens
constexpr int AM_D = 64;          // head dim
constexpr int AM_S = 72;          // LDS row stride (elements): 144 B -> conflict-free ds_read_b128 fragments

// stage one [T x 64] bf16 head slice into a wave-private LDS tile (rows >= T zero-filled).  All eight 16-byte loads
// are issued back to back on a clamped row index (a per-load bounds branch would make hipcc wait for each load in turn);
// the zero fill is a select after the loads.
VC_DEV void am_stage(vc_bf16* tile, const vc_bf16* g, long ld, int T, int lane) {
    vc_u32x4 v[AM_T / 8];
#pragma unroll
    for (int it = 0; it < AM_T / 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        const int rr = row < T ? row : T - 1;
        v[it] = *reinterpret_cast<const vc_u32x4*>(g + (long)rr * ld + c);
    }
#pragma unroll
    for (int it = 0; it < AM_T / 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        vc_u32x4 w = v[it];
        if (row >= T) { w.x = 0u; w.y = 0u; w.z = 0u; w.w = 0u; }
        *reinterpret_cast<vc_u32x4*>(tile + row * AM_S + c) = w;
    }
}
// direct fragment: row = row0 + (lane&31), 8 consecutive d at ks*16 + 8*(lane>>5)
VC_DEV vc_s16x8 am_frag(const vc_bf16* tile, int row0, int ks, int lane) {
    return *reinterpret_cast<const vc_s16x8*>(tile + (row0 + (lane & 31)) * AM_S + ks * 16 + (lane >> 5) * 8);
}
// transposed fragment for the "token" contraction: column n0 + (lane&31), token rows k0 + 4h + {0..3} and + 8 + {0..3}
VC_DEV vc_s16x8 am_frag_tr(const vc_bf16* tile, int k0, int n0, int lane) {
    const int i = lane & 15;
    const vc_bf16* p = tile + (k0 + 4 * (lane >> 5) + (i >> 2)) * AM_S + n0 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
    const vc_s16x4 lo = vc_ds_read_tr16(p), hi = vc_ds_read_tr16(p + 8 * AM_S);
    vc_s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// accumulator registers 8s..8s+7 of a 32x32 tile -> bf16 A/B fragment (k-slots = rows 4h+16s+{0..3, 8..11})
VC_DEV vc_s16x8 am_pack(const vc_f32x16& a, int s) {
    vc_s16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)vc_f32_to_bf16(a[8 * s + j]).bits;
    return r;
}
// same, with the dropout keep-multipliers of tile (ti, tj) applied while packing (P' = P * mask never materialised in fp32)
VC_DEV vc_s16x8 am_pack_keep(const vc_f32x16& a, int s, uint64_t keep, int ti, int tj, float scale) {
    vc_s16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float m = ((keep >> ((ti * 2 + tj) * 16 + 8 * s + j)) & 1) ? scale : 0.0f;
        r[j] = (short)vc_f32_to_bf16(a[8 * s + j] * m).bits;
    }
    return r;
}
VC_DEV int am_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }   // row inside a 32x32 D tile

// Workgroup -> (image n, head h) of the ViT kernels, XCD-aware (r04): hardware deals workgroups round-robin to the 8 XCDs (block b runs on XCD b % 8), so
// with the plain b -> (b / H, b % H) order the 16 heads of one frame — sixteen adjacent 128-byte pieces of the same q / k / v rows — were fetched by eight different L2s,
// each pulling single cache lines out of every 6 KiB row.  Here XCD x takes whole frames (frames 8 j + x): its L2 sees every row as three contiguous 2 KiB runs.
// The frames past the last multiple of 8 keep the plain order.  (Pure placement: no effect on results.)
VC_DEV void am_block_to_frame_head(int b, int B, int H, long& n, int& h) {
    const int full = (B >> 3) << 3;                       // frames covered by whole groups of 8
    if (b < full * H) { const int xcd = b & 7, idx = b >> 3; n = (long)(idx / H) * 8 + xcd; h = idx % H; }
    else { n = b / H; h = b % H; }
}

VC_DEV void am_zero(vc_f32x16 (&a)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[i][j][r] = 0.f;
}
// acc[i][j] += X[rows i*32..][d] * Y[rows j*32..][d]^T over d = 0..63 (both direct fragments)
VC_DEV void am_mm_nt(vc_f32x16 (&acc)[2][2], const vc_bf16* X, const vc_bf16* Y, int lane) {
#pragma unroll
    for (int ks = 0; ks < AM_D / 16; ++ks) {
        vc_s16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a[i] = am_frag(X, i * 32, ks, lane); b[i] = am_frag(Y, i * 32, ks, lane); }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = vc_mfma_32x32x16_bf16(a[i], b[j], acc[i][j]);
    }
}
// out[ot][dt] += sum over token tiles tt: W[tt][ot]^T-style contraction: A = pack(W[tt][ot]) (lane = out row), B = Ytile rows (tokens) via tr
// W is indexed W[tt][ot] (accumulator grid whose *register rows* are the contracted tokens and lane column the output row)
VC_DEV void am_mm_tok(vc_f32x16 (&out)[2][2], const vc_f32x16 (&W)[2][2], const vc_bf16* Y, int lane) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            vc_s16x8 a[2], b[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) { a[o] = am_pack(W[tt][o], s); b[o] = am_frag_tr(Y, tt * 32 + 16 * s, o * 32, lane); }
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int d = 0; d < 2; ++d) out[o][d] = vc_mfma_32x32x16_bf16(a[o], b[d], out[o][d]);
        }
}
template <bool DROP>
VC_DEV void am_mm_tok_keep(vc_f32x16 (&out)[2][2], const vc_f32x16 (&W)[2][2], const vc_bf16* Y, int lane, uint64_t keep, float scale) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            vc_s16x8 a[2], b[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                a[o] = DROP ? am_pack_keep(W[tt][o], s, keep, tt, o, scale) : am_pack(W[tt][o], s);
                b[o] = am_frag_tr(Y, tt * 32 + 16 * s, o * 32, lane);
            }
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int d = 0; d < 2; ++d) out[o][d] = vc_mfma_32x32x16_bf16(a[o], b[d], out[o][d]);
        }
}
// store a [64 x 64] result grid (rows = tokens, lane column = d) as bf16, rows < T only
VC_DEV void am_store(vc_bf16* g, long ld, const vc_f32x16 (&acc)[2][2], int T, int lane, float mul) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = o * 32 + am_row(r, lane);
                if (row < T) g[(long)row * ld + d * 32 + (lane & 31)] = vc_f32_to_bf16(acc[o][d][r] * mul);
            }
}

// Keep-bits of the attention-probability dropout for this lane's 64 accumulator elements (2x2 tiles x 16 registers),
// packed into one 64-bit word so the mask costs 2 registers instead of 64 live multipliers.
// bit (ti*2 + tj)*16 + r  <->  element (row = ti*32 + am_row(r), lane column = tj*32 + (lane&31)) of a grid whose rows are
// keys and columns queries when QCOL, else rows = queries and columns = keys.   idx = base + query*T + key.
// (q0, k0: global offsets of the 64x64 block, multiples of 64.)  One hash serves two consecutive elements (vc_rt.h): with T even
// an aligned key pair of one query shares it, so
//   QCOL  (registers walk the keys)   : accumulator registers r = 2j, 2j+1 hold keys k, k+1 — one hash per register pair;
//   !QCOL (registers walk the queries): lanes l, l^1 hold keys k, k+1 of the same queries — each computes the hashes of every other
//                                        register row and they swap (one xor-1 shuffle per hash);
// odd T (never the case for the canonical 50-token ViT / even horizons) falls back to one hash per element.
template <bool QCOL>
VC_DEV uint64_t am_keep_bits_g(const vc_drop& d, uint32_t base, int T, int q0, int k0, int lane) {
    uint32_t lo = 0, hi = 0;                  // tiles (0,0),(0,1) -> lo ; (1,0),(1,1) -> hi   (rolled loops: keep register pressure flat)
    if ((T & 1) == 0) {
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ti = t >> 1, tj = t & 1;
                const int col = tj * 32 + (lane & 31);
                uint32_t b0, b1;              // keep bits of registers 2j and 2j + 1
                if (QCOL) {
                    const int key = k0 + ti * 32 + am_row(2 * j, lane);                       // even; register 2j+1 holds key + 1
                    const uint32_t h = vc_drop_hash(d, (base + (uint32_t)((q0 + col) * T + key)) >> 1);
                    b0 = vc_drop_keep_lo(d, h) ? 1u : 0u; b1 = vc_drop_keep_hi(d, h) ? 1u : 0u;
                } else {
                    const int par = lane & 1;                                                 // this lane hashes register row 2j + par
                    const int query = q0 + ti * 32 + am_row(2 * j + par, lane);
                    const uint32_t mine = vc_drop_hash(d, (base + (uint32_t)(query * T + k0 + (col & ~1))) >> 1);
                    const uint32_t other = (uint32_t)vc_shfl_xor((int)mine, 1);               // partner lane: same key pair, row 2j + 1 - par
                    const uint32_t h0 = par ? other : mine, h1 = par ? mine : other;          // hashes of rows 2j, 2j + 1
                    b0 = ((col & 1) ? vc_drop_keep_hi(d, h0) : vc_drop_keep_lo(d, h0)) ? 1u : 0u;
                    b1 = ((col & 1) ? vc_drop_keep_hi(d, h1) : vc_drop_keep_lo(d, h1)) ? 1u : 0u;
                }
                const uint32_t two = (b0 | (b1 << 1)) << (tj * 16 + 2 * j);
                if (ti == 0) lo |= two; else hi |= two;
            }
        }
    } else {
#pragma unroll 1
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ti = t >> 1, tj = t & 1;
                const int row = ti * 32 + am_row(r, lane), col = tj * 32 + (lane & 31);
                const int query = q0 + (QCOL ? col : row), key = k0 + (QCOL ? row : col);
                const uint32_t bit = vc_drop_keep(d, base + (uint32_t)(query * T + key)) ? 1u : 0u;
                if (ti == 0) lo |= bit << (tj * 16 + r); else hi |= bit << (tj * 16 + r);
            }
        }
    }
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
template <bool QCOL>
VC_DEV uint64_t am_keep_bits(const vc_drop& d, uint32_t base, int T, int lane) { return am_keep_bits_g<QCOL>(d, base, T, 0, 0, lane); }
VC_DEV float am_keep(uint64_t bits, int ti, int tj, int r, float scale) { return ((bits >> ((ti * 2 + tj) * 16 + r)) & 1) ? scale : 0.0f; }

template <bool DROP>
VC_KERNEL __launch_bounds__(64) void attn_vit_fwd_mfma_kernel(AttnParams p) {
    // two LDS tiles (18 KB): V takes Q's tile once S is in registers — 8 instead of 5 single-wave workgroups per CU
    VC_SHARED __attribute__((aligned(16))) vc_bf16 tiles[2][AM_T * AM_S];
    const int lane = threadIdx.x;
    const int h = blockIdx.x % p.H; const long n = blockIdx.x / p.H;
    const int T = p.Tq;
    const long rowq = n * T;
    am_stage(tiles[0], (const vc_bf16*)p.q + rowq * p.ldq + h * AM_D, p.ldq, T, lane);
    am_stage(tiles[1], (const vc_bf16*)p.k + rowq * p.ldk + h * AM_D, p.ldk, T, lane);
    vc_wave_barrier();
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    uint64_t keep = 0;
    if (DROP) keep = am_keep_bits<true>(p.drop, dbase0, T, lane);
    vc_f32x16 st[2][2];                       // S^T: [key tile][query tile], lane column = query
    am_zero(st);
    am_mm_nt(st, tiles[1], tiles[0], lane);
    vc_wave_barrier();                        // Q's fragments are consumed: its tile now receives V (the load overlaps the softmax)
    am_stage(tiles[0], (const vc_bf16*)p.v + rowq * p.ldv + h * AM_D, p.ldv, T, lane);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + am_row(r, lane);
                const float s = (key < T) ? st[kt][qt][r] * p.scale : -INFINITY;
                st[kt][qt][r] = s; m = fmaxf(m, s);
            }
        m = fmaxf(m, vc_shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = expf(st[kt][qt][r] - m); st[kt][qt][r] = e; l += e; }
        l += vc_shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int query = qt * 32 + (lane & 31);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][qt][r] *= DROP ? inv * am_keep(keep, kt, qt, r, p.drop.scale) : inv;
        if (p.lse && lane < 32 && query < T) p.lse[(n * p.H + h) * T + query] = m + logf(l);
    }
    vc_wave_barrier();
    vc_f32x16 o[2][2];
    am_zero(o);
    am_mm_tok(o, st, tiles[0], lane);        // O[query][d] = sum_key P[query][key] V[key][d]
    am_store((vc_bf16*)p.o + rowq * p.ldo + h * AM_D, p.ldo, o, T, lane, 1.0f);
}

// stage with NT threads cooperating (backward: 2 waves share the tiles)
template <int NT>
VC_DEV void am_stage_nt(vc_bf16* tile, const vc_bf16* g, long ld, int T, int tid) {
    constexpr int N = AM_T * 8 / NT;
    vc_u32x4 v[N];
#pragma unroll
    for (int it = 0; it < N; ++it) {
        const int c = tid + NT * it, row = c >> 3, col = (c & 7) * 8;
        v[it] = *reinterpret_cast<const vc_u32x4*>(g + (long)(row < T ? row : T - 1) * ld + col);
    }
#pragma unroll
    for (int it = 0; it < N; ++it) {
        const int c = tid + NT * it, row = c >> 3, col = (c & 7) * 8;
        vc_u32x4 w = v[it];
        if (row >= T) { w.x = 0u; w.y = 0u; w.z = 0u; w.w = 0u; }
        *reinterpret_cast<vc_u32x4*>(tile + row * AM_S + col) = w;
    }
}

// Backward: one 2-wave block per (frame, head).  Both waves share the staged Q/K/V/dO tiles; wave 0 owns the
// "lane = query" orientation (D_i, dQ), wave 1 the "lane = key" orientation (dV, dK).  Wave 1 runs its two S / dP
// MFMA grids while wave 0 produces D_i; the hand-off is the block barrier.  Halving the per-wave accumulator set
// lifts occupancy from 1 to 2 waves per SIMD (8 waves per CU, LDS-limited at 37 KB per block).
template <bool DROP>
VC_DEV void attn_vit_bwd_mfma_body(const AttnParams& p) {
    VC_SHARED __attribute__((aligned(16))) vc_bf16 tiles[4][AM_T * AM_S];     // Q, K, V, dO
    VC_SHARED float lse_s[AM_T];
    VC_SHARED float del_s[AM_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x % p.H; const long n = blockIdx.x / p.H;
    const int T = p.Tq;
    const long rowq = n * T;
    am_stage_nt<128>(tiles[0], (const vc_bf16*)p.q + rowq * p.ldq + h * AM_D, p.ldq, T, tid);
    am_stage_nt<128>(tiles[1], (const vc_bf16*)p.k + rowq * p.ldk + h * AM_D, p.ldk, T, tid);
    am_stage_nt<128>(tiles[2], (const vc_bf16*)p.v + rowq * p.ldv + h * AM_D, p.ldv, T, tid);
    am_stage_nt<128>(tiles[3], (const vc_bf16*)p.dout + rowq * p.lddo + h * AM_D, p.lddo, T, tid);
    if (tid < AM_T) lse_s[tid] = (tid < T) ? p.lse[(n * p.H + h) * T + tid] : 0.f;
    vc_sync();
    const vc_bf16 *Qs = tiles[0], *Ks = tiles[1], *Vs = tiles[2], *dOs = tiles[3];

    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    if (wave == 0) {   // ---------------- lane = query:  D_i, dQ
        uint64_t keep = 0;
        if (DROP) keep = am_keep_bits<true>(p.drop, dbase0, T, lane);
        vc_f32x16 st[2][2], dpt[2][2];
        am_zero(st); am_zero(dpt);
        am_mm_nt(st, Ks, Qs, lane);          // S^T[key][query]
        am_mm_nt(dpt, Vs, dOs, lane);        // dP^T[key][query] = sum_d V[key][d] dO[query][d]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int query = qt * 32 + (lane & 31);
            const float lse = lse_s[query];
            float dsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + am_row(r, lane);
                    const bool ok = key < T && query < T;
                    const float pr = ok ? expf(st[kt][qt][r] * p.scale - lse) : 0.f;
                    if (DROP) dpt[kt][qt][r] *= am_keep(keep, kt, qt, r, p.drop.scale);                    // dP = dP' * mask
                    st[kt][qt][r] = pr; dsum += pr * dpt[kt][qt][r];
                }
            dsum += vc_shfl_xor(dsum, 32);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][qt][r] = st[kt][qt][r] * (dpt[kt][qt][r] - dsum);   // dS^T (scale folded into the store)
            if (lane < 32) del_s[query] = dsum;
        }
        vc_sync();                           // publish D_i to wave 1
        vc_f32x16 dq[2][2];
        am_zero(dq);
        am_mm_tok(dq, st, Ks, lane);         // dQ[query][d] = sum_key dS[query][key] K[key][d]
        am_store((vc_bf16*)p.dq + rowq * p.lddq + h * AM_D, p.lddq, dq, T, lane, p.scale);
    } else {           // ---------------- lane = key:  dV, dK
        uint64_t keep = 0;
        if (DROP) keep = am_keep_bits<false>(p.drop, dbase0, T, lane);
        // ordered so that at most two 64-register grids are live: S -> P ; dV = P'^T dO ; dP ; dS ; dK = dS^T Q
        vc_f32x16 sn[2][2];
        am_zero(sn);
        am_mm_nt(sn, Qs, Ks, lane);          // S[query][key]   (grid [query tile][key tile], lane column = key)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = kt * 32 + (lane & 31);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int query = qt * 32 + am_row(r, lane);
                    sn[qt][kt][r] = (key < T && query < T) ? expf(sn[qt][kt][r] * p.scale - lse_s[query]) : 0.f;   // P
                }
        }
        {
            vc_f32x16 acc[2][2];
            am_zero(acc);
            am_mm_tok_keep<DROP>(acc, sn, dOs, lane, keep, p.drop.scale);     // dV[key][d] = sum_query P'[query][key] dO[query][d]
            am_store((vc_bf16*)p.dv + rowq * p.lddv + h * AM_D, p.lddv, acc, T, lane, 1.0f);
        }
        vc_f32x16 dp[2][2];
        am_zero(dp);
        am_mm_nt(dp, dOs, Vs, lane);         // dP'[query][key]
        vc_sync();                           // D_i from wave 0
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int query = qt * 32 + am_row(r, lane);
                    const float ms = DROP ? am_keep(keep, qt, kt, r, p.drop.scale) : 1.0f;
                    dp[qt][kt][r] = sn[qt][kt][r] * (dp[qt][kt][r] * ms - del_s[query]);                          // dS
                }
        vc_f32x16 acc[2][2];
        am_zero(acc);
        am_mm_tok(acc, dp, Qs, lane);        // dK[key][d] = sum_query dS[query][key] Q[query][d]
        am_store((vc_bf16*)p.dk + rowq * p.lddk + h * AM_D, p.lddk, acc, T, lane, p.scale);
    }
}

// ---- r02: the same backward on FOUR waves per (frame, head): wave = (orientation, 32-column tile).
// r01's two waves carried 2x2-tile grids (64 accumulator registers each, three live): 256 registers -> 2 waves per SIMD, and the
// kernel ran at the pace of one wave's ~7 k-instruction dependency chain (22 us per workgroup, 3 TB/s).  Here every wave owns ONE
// 32-wide column tile of its orientation (queries for dQ, keys for dV / dK): grids are 2 x 1 tiles, ~128 registers, 4 waves per
// SIMD, half the work per wave — and the token-contraction products are issued with swapped operands (D^T), so a lane ends up with
// a token ROW and 4 consecutive head-dim columns per accumulator quad: 8-byte row stores, 8 per output tile instead of 32 scalars.
template <bool QCOL>
VC_DEV uint32_t am_keep_bits1(const vc_drop& d, uint32_t base, int T, int tj, int lane) {
    // bit ti*16 + r  <->  element (row = ti*32 + am_row(r), lane column = tj*32 + (lane&31)); rows = keys / column = query when QCOL,
    // else rows = queries / column = key.  Same pairing of one hash per two consecutive keys as am_keep_bits_g.
    uint32_t bits = 0;
    const int col = tj * 32 + (lane & 31);
    if ((T & 1) == 0) {
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                uint32_t b0, b1;
                if (QCOL) {
                    const int key = ti * 32 + am_row(2 * j, lane);
                    const uint32_t h = vc_drop_hash(d, (base + (uint32_t)(col * T + key)) >> 1);
                    b0 = vc_drop_keep_lo(d, h) ? 1u : 0u; b1 = vc_drop_keep_hi(d, h) ? 1u : 0u;
                } else {
                    const int par = lane & 1;
                    const int query = ti * 32 + am_row(2 * j + par, lane);
                    const uint32_t mine = vc_drop_hash(d, (base + (uint32_t)(query * T + (col & ~1))) >> 1);
                    const uint32_t other = (uint32_t)vc_shfl_xor((int)mine, 1);
                    const uint32_t h0 = par ? other : mine, h1 = par ? mine : other;
                    b0 = ((col & 1) ? vc_drop_keep_hi(d, h0) : vc_drop_keep_lo(d, h0)) ? 1u : 0u;
                    b1 = ((col & 1) ? vc_drop_keep_hi(d, h1) : vc_drop_keep_lo(d, h1)) ? 1u : 0u;
                }
                bits |= (b0 | (b1 << 1)) << (ti * 16 + 2 * j);
            }
        }
    } else {
#pragma unroll 1
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const int row = ti * 32 + am_row(r, lane);
                const int query = QCOL ? col : row, key = QCOL ? row : col;
                bits |= (vc_drop_keep(d, base + (uint32_t)(query * T + key)) ? 1u : 0u) << (ti * 16 + r);
            }
    }
    return bits;
}
VC_DEV float am_keep1(uint32_t bits, int ti, int r, float scale) { return ((bits >> (ti * 16 + r)) & 1) ? scale : 0.0f; }
VC_DEV vc_s16x8 am_pack_keep1(const vc_f32x16& a, int s, uint32_t keep, int ti, float scale) {
    vc_s16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)vc_f32_to_bf16(a[8 * s + j] * (((keep >> (ti * 16 + 8 * s + j)) & 1) ? scale : 0.0f)).bits;
    return r;
}
// acc[i] += X[rows i*32..][d] * Y[rows t*32..][d]^T   (registers = X rows, lane column = Y row of tile t)
VC_DEV void am_mm_nt1(vc_f32x16 (&acc)[2], const vc_bf16* X, const vc_bf16* Y, int t, int lane) {
#pragma unroll
    for (int ks = 0; ks < AM_D / 16; ++ks) {
        const vc_s16x8 b = am_frag(Y, t * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = vc_mfma_32x32x16_bf16(am_frag(X, i * 32, ks, lane), b, acc[i]);
    }
}
// out[dt][d-row (registers)][token (lane)] += sum over the 64 contracted tokens (register rows of W[tt], tt = 0, 1) of W * Y[token][d]
template <bool DROP>
VC_DEV void am_mm_tok1(vc_f32x16 (&out)[2], const vc_f32x16 (&W)[2], const vc_bf16* Y, int lane, uint32_t keep, float scale) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const vc_s16x8 b = DROP ? am_pack_keep1(W[tt], s, keep, tt, scale) : am_pack(W[tt], s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) out[dt] = vc_mfma_32x32x16_bf16(am_frag_tr(Y, tt * 32 + 16 * s, dt * 32, lane), b, out[dt]);
        }
}
// rows t*32 .. t*32+31 (< T) of a [token][64] result whose registers walk the head dim (lane = token row, 4 consecutive columns per
// accumulator quad): transposed through a wave-private 32 x AM_S staging tile (8-byte LDS writes), then written as whole 128-byte rows —
// 16 bytes per lane, 8 full lines per store instruction.  (Storing the quads straight from the registers — 8 bytes per lane at 32
// different rows — was measured first: 16x the write transactions, the forward got slower.)
VC_DEV void am_store_rows(vc_bf16* stage, vc_bf16* g, long ld, const vc_f32x16 (&acc)[2], int t, int T, int lane, float mul) {
    vc_bf16* w = stage + (lane & 31) * AM_S + 4 * (lane >> 5);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            vc_u32x2 q;
            q.x = vc_pack_bf16x2(acc[dt][4 * gq] * mul, acc[dt][4 * gq + 1] * mul); q.y = vc_pack_bf16x2(acc[dt][4 * gq + 2] * mul, acc[dt][4 * gq + 3] * mul);
            *reinterpret_cast<vc_u32x2*>(w + dt * 32 + 8 * gq) = q;
        }
    vc_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        const vc_u32x4 v = *reinterpret_cast<const vc_u32x4*>(stage + row * AM_S + c);
        if (t * 32 + row < T) *reinterpret_cast<vc_u32x4*>(g + (long)(t * 32 + row) * ld + c) = v;
    }
    vc_wave_barrier();
}
VC_DEV void am_zero1(vc_f32x16 (&a)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[i][r] = 0.f;
}

template <bool DROP>
VC_DEV void attn_vit_bwd4_body(const AttnParams& p) {
    VC_SHARED __attribute__((aligned(16))) vc_bf16 tiles[4][AM_T * AM_S];     // Q, K, V, dO
    VC_SHARED float lse_s[AM_T];
    VC_SHARED float del_s[AM_T];
#if defined(VCAD_AB) && !defined(VC_EMU)
    VC_SHARED uint32_t pf_sink[2 * 64];                                        // landing place of the L2 warm-up DMAs (never read)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = vc_uniform(tid >> 6);
    int h; long n; am_block_to_frame_head((int)blockIdx.x, p.B, p.H, n, h);
    const int T = p.Tq;
    const long rowq = n * T;
    am_stage_nt<256>(tiles[0], (const vc_bf16*)p.q + rowq * p.ldq + h * AM_D, p.ldq, T, tid);
    am_stage_nt<256>(tiles[1], (const vc_bf16*)p.k + rowq * p.ldk + h * AM_D, p.ldk, T, tid);
    am_stage_nt<256>(tiles[2], (const vc_bf16*)p.v + rowq * p.ldv + h * AM_D, p.ldv, T, tid);
    am_stage_nt<256>(tiles[3], (const vc_bf16*)p.dout + rowq * p.lddo + h * AM_D, p.lddo, T, tid);
    if (tid < AM_T) lse_s[tid] = (tid < T) ? p.lse[(n * p.H + h) * T + tid] : 0.f;
    vc_sync();
    const vc_bf16 *Qs = tiles[0], *Ks = tiles[1], *Vs = tiles[2], *dOs = tiles[3];
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    const int t = wave & 1;
    if (wave < 2) {    // ---------------- lane = query of tile t:  D_i, dQ
        uint32_t keep = 0;
        if (DROP) keep = am_keep_bits1<true>(p.drop, dbase0, T, t, lane);
#if defined(VCAD_AB) && !defined(VC_EMU)
        // r06 experiment, A/B build only (vcad_debug_attn_prefetch; profiles/r06_attn_prefetch_ab.txt: SLOWER, +0.2 % / +0.6 % / +0.5 % on the step at 16 / 64 / 128 frames ahead).
        // Hypothesis: a workgroup is one dependent chain — tiles in (a memory round trip), ~5 us of matrix / VALU work, gradients out — and a CU holds four of them, so on
        // average little more than one workgroup per CU has loads in flight (~30 KB against the ~54 KB that 5.5 TB/s x the loaded latency asks of a CU); the two query
        // waves therefore warm the L2 for the (frame + pf_frames, head) pair a later workgroup of the SAME XCD will stage (am_block_to_frame_head: XCD x owns frames
        // 8 j + x): 200 lines, one 4-byte DMA per line (wave 0: Q and K rows, wave 1: V and dO rows), issued where the next vmcnt wait of these waves is ~500
        // instructions away (vmcnt retires in order).  The kernel is not waiting for bytes in flight.
        if (p.pf_frames > 0 && n + p.pf_frames < p.B) {
            const int row = lane < T ? lane : T - 1;
            const long r2 = (n + p.pf_frames) * T + row;
            const vc_bf16* s0 = wave == 0 ? (const vc_bf16*)p.q + r2 * p.ldq : (const vc_bf16*)p.v + r2 * p.ldv;
            const vc_bf16* s1 = wave == 0 ? (const vc_bf16*)p.k + r2 * p.ldk : (const vc_bf16*)p.dout + r2 * p.lddo;
            vc_prefetch_line(s0 + h * AM_D, pf_sink + wave * 64);
            vc_prefetch_line(s1 + h * AM_D, pf_sink + wave * 64);
        }
#endif
        vc_f32x16 st[2], dpt[2];
        am_zero1(st); am_zero1(dpt);
        am_mm_nt1(st, Ks, Qs, t, lane);      // S^T[key][query]
        am_mm_nt1(dpt, Vs, dOs, t, lane);    // dP^T[key][query]
        const int query = t * 32 + (lane & 31);
        const float lse = lse_s[query];
        float dsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + am_row(r, lane);
                const bool ok = key < T && query < T;
                const float pr = ok ? vc_expf_fast(st[kt][r] * p.scale - lse) : 0.f;
                if (DROP) dpt[kt][r] *= am_keep1(keep, kt, r, p.drop.scale);
                st[kt][r] = pr; dsum += pr * dpt[kt][r];
            }
        dsum += vc_shfl_xor(dsum, 32);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = st[kt][r] * (dpt[kt][r] - dsum);      // dS^T (scale folded into the store)
        if (lane < 32) del_s[query] = dsum;
        vc_sync();                           // publish D_i to the key waves
        vc_f32x16 dq[2];
        am_zero1(dq);
        am_mm_tok1<false>(dq, st, Ks, lane, 0u, 1.0f);          // dQ[query][d] = sum_key dS[query][key] K[key][d]
#if defined(VCAD_AB) && !defined(VC_EMU)
        vc_wait_vmcnt<0>();                  // (the warm-up DMAs: no LDS write of this wave may be pending when the workgroup's LDS is handed on)
#endif
        am_store_rows(tiles[2] + t * 32 * AM_S, (vc_bf16*)p.dq + rowq * p.lddq + h * AM_D, p.lddq, dq, t, T, lane, p.scale);   // (V is dead after the barrier)
    } else {           // ---------------- lane = key of tile t:  dV, dK
        uint32_t keep = 0;
        if (DROP) keep = am_keep_bits1<false>(p.drop, dbase0, T, t, lane);
        vc_f32x16 sn[2];
        am_zero1(sn);
        am_mm_nt1(sn, Qs, Ks, t, lane);      // S[query][key]
        const int key = t * 32 + (lane & 31);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int query = qt * 32 + am_row(r, lane);
                sn[qt][r] = (key < T && query < T) ? vc_expf_fast(sn[qt][r] * p.scale - lse_s[query]) : 0.f;       // P
            }
        vc_f32x16 dv[2];
        am_zero1(dv);
        am_mm_tok1<DROP>(dv, sn, dOs, lane, keep, p.drop.scale);         // dV[key][d] = sum_query P'[query][key] dO[query][d]
        vc_f32x16 dp[2];
        am_zero1(dp);
        am_mm_nt1(dp, dOs, Vs, t, lane);     // dP'[query][key]
        vc_sync();                           // D_i from the query waves; nobody reads V / dO any more: their tiles become the store staging
        vc_bf16* stage = tiles[3] + t * 32 * AM_S;
        am_store_rows(stage, (vc_bf16*)p.dv + rowq * p.lddv + h * AM_D, p.lddv, dv, t, T, lane, 1.0f);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int query = qt * 32 + am_row(r, lane);
                const float ms = DROP ? am_keep1(keep, qt, r, p.drop.scale) : 1.0f;
                dp[qt][r] = sn[qt][r] * (dp[qt][r] * ms - del_s[query]);                                    // dS
            }
        vc_f32x16 dk[2];
        am_zero1(dk);
        am_mm_tok1<false>(dk, dp, Qs, lane, 0u, 1.0f);          // dK[key][d] = sum_query dS[query][key] Q[query][d]
        am_store_rows(stage, (vc_bf16*)p.dk + rowq * p.lddk + h * AM_D, p.lddk, dk, t, T, lane, p.scale);
    }
}
VC_KERNEL __launch_bounds__(256, 4) void attn_vit_bwd4_kernel_eval(AttnParams p) { attn_vit_bwd4_body<false>(p); }
VC_KERNEL __launch_bounds__(256, 4) void attn_vit_bwd4_kernel_drop(AttnParams p) { attn_vit_bwd4_body<true>(p); }

// ---- r02 forward: TWO waves per (frame, head), one 32-query tile each (grids of 2 x 1 tiles: 4 waves per SIMD instead of 2), O = P V
// issued with swapped operands so a lane owns a query row and stores 8-byte pieces of it (see the four-wave backward below).
template <bool DROP>
VC_KERNEL __launch_bounds__(128, 4) void attn_vit_fwd2_kernel(AttnParams p) {
    VC_SHARED __attribute__((aligned(16))) vc_bf16 tiles[2][AM_T * AM_S];      // Q then V, K
    const int tid = threadIdx.x, lane = tid & 63, t = vc_uniform(tid >> 6);
    int h; long n; am_block_to_frame_head((int)blockIdx.x, p.B, p.H, n, h);
    const int T = p.Tq;
    const long rowq = n * T;
    am_stage_nt<128>(tiles[0], (const vc_bf16*)p.q + rowq * p.ldq + h * AM_D, p.ldq, T, tid);
    am_stage_nt<128>(tiles[1], (const vc_bf16*)p.k + rowq * p.ldk + h * AM_D, p.ldk, T, tid);
    // V is requested NOW (into registers) and parked in Q's tile once S is done: one memory round trip per workgroup instead of two
    vc_u32x4 vreg[AM_T * 8 / 128];
    {
        const vc_bf16* gv = (const vc_bf16*)p.v + rowq * p.ldv + h * AM_D;
#pragma unroll
        for (int it = 0; it < AM_T * 8 / 128; ++it) {
            const int c = tid + 128 * it, row = c >> 3, col = (c & 7) * 8;
            vreg[it] = *reinterpret_cast<const vc_u32x4*>(gv + (long)(row < T ? row : T - 1) * p.ldv + col);
        }
    }
    vc_sync();
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    uint32_t keep = 0;
    if (DROP) keep = am_keep_bits1<true>(p.drop, dbase0, T, t, lane);
    vc_f32x16 st[2];                          // S^T[key tile][query tile t], lane column = query
    am_zero1(st);
    am_mm_nt1(st, tiles[1], tiles[0], t, lane);
    vc_sync();                                // both waves are done with Q: its tile now receives V
#pragma unroll
    for (int it = 0; it < AM_T * 8 / 128; ++it) {
        const int c = tid + 128 * it, row = c >> 3, col = (c & 7) * 8;
        vc_u32x4 w = vreg[it];
        if (row >= T) { w.x = 0u; w.y = 0u; w.z = 0u; w.w = 0u; }
        *reinterpret_cast<vc_u32x4*>(tiles[0] + row * AM_S + col) = w;
    }
    {
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + am_row(r, lane);
                const float sv = (key < T) ? st[kt][r] * p.scale : -INFINITY;
                st[kt][r] = sv; m = fmaxf(m, sv);
            }
        m = fmaxf(m, vc_shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = vc_expf_fast(st[kt][r] - m); st[kt][r] = e; l += e; }
        l += vc_shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int query = t * 32 + (lane & 31);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] *= DROP ? inv * am_keep1(keep, kt, r, p.drop.scale) : inv;
        if (p.lse && lane < 32 && query < T) p.lse[(n * p.H + h) * T + query] = m + logf(l);
    }
    vc_sync();                                // V has landed
    vc_f32x16 o[2];
    am_zero1(o);
    am_mm_tok1<false>(o, st, tiles[0], lane, 0u, 1.0f);        // O[query][d] = sum_key P[query][key] V[key][d]
    am_store_rows(tiles[1] + t * 32 * AM_S, (vc_bf16*)p.o + rowq * p.ldo + h * AM_D, p.ldo, o, t, T, lane, 1.0f);       // (K is dead since the first barrier)
}

// eval / p = 0: capped at 256 registers -> 2 waves per SIMD.  Train mode (mask bits + masked packing) wants 332; capping it at 256
// spills 38 registers to scratch but doubles the waves per SIMD, which wins (measured: 487 -> ~370 us per call; ATTN_BWD_DROP_WAVES=1
// restores the uncapped build).
#ifndef ATTN_BWD_DROP_WAVES
#define ATTN_BWD_DROP_WAVES 2
#endif
VC_KERNEL __launch_bounds__(128, 2) void attn_vit_bwd_mfma_kernel_eval(AttnParams p) { attn_vit_bwd_mfma_body<false>(p); }
VC_KERNEL __launch_bounds__(128, ATTN_BWD_DROP_WAVES) void attn_vit_bwd_mfma_kernel_drop(AttnParams p) { attn_vit_bwd_mfma_body<true>(p); }

// ====================================================================================================================
// Decoder attention on the matrix cores: one (clip, head) per workgroup, T <= 64 steps, head dim = NCH chunks of 64.
//
// Same orientation tricks as the ViT kernels above; the head dimension (256 in the canonical model) is walked in 64-wide
// chunks that are re-staged through the same small LDS tiles: the score grids accumulate over the chunks, the outputs are
// produced one chunk at a time from the probability / dS grids that stay in registers.  The mask is the reference's
// (causal + window band, model/autoregressive_transformer.py:180-188): key j visible to query i iff i - window < j <= i.
// The wave-per-row kernels in attn.h remain the path for fp32 mode beyond 64 steps.  Softmax exponentials are v_exp_f32 (vc_expf_fast, r04; r01-r03: libm
// expf, ~20 instructions each — 64 per lane and block pair): exp(0) = 1 exactly either way, which is all the single-visible-key rule needs, and
// the forward's lse and the backward's recompute use the same function.
// ====================================================================================================================
VC_DEV bool am_visible(int query, int key, int T, int window) { return key < T && key <= query && key > query - window; }

template <bool DROP, int NCH>
VC_KERNEL __launch_bounds__(64) void attn_dec_fwd_mfma_kernel(AttnParams p) {
    VC_SHARED __attribute__((aligned(16))) vc_bf16 tiles[2][AM_T * AM_S];
    const int lane = threadIdx.x;
    const int h = blockIdx.x % p.H; const long n = blockIdx.x / p.H;
    const int T = p.Tq;
    const long rowq = n * T;
    const int hd = h * AM_D * NCH;
    vc_f32x16 st[2][2];                       // S^T: [key tile][query tile], lane column = query
    am_zero(st);
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        vc_wave_barrier();                    // the previous chunk's fragment reads are done before its tiles are overwritten
        am_stage(tiles[0], (const vc_bf16*)p.q + rowq * p.ldq + hd + c * AM_D, p.ldq, T, lane);
        am_stage(tiles[1], (const vc_bf16*)p.k + rowq * p.ldk + hd + c * AM_D, p.ldk, T, lane);
        vc_wave_barrier();
        am_mm_nt(st, tiles[1], tiles[0], lane);
    }
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    uint64_t keep = 0;
    if (DROP) keep = am_keep_bits<true>(p.drop, dbase0, T, lane);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int query = qt * 32 + (lane & 31);
        const int qm = query < T ? query : T - 1;              // padding columns: keep the row finite (never stored)
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + am_row(r, lane);
                const float sc = am_visible(qm, key, T, p.window) ? st[kt][qt][r] * p.scale : -INFINITY;
                st[kt][qt][r] = sc; m = fmaxf(m, sc);
            }
        m = fmaxf(m, vc_shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = vc_expf_fast(st[kt][qt][r] - m); st[kt][qt][r] = e; l += e; }
        l += vc_shfl_xor(l, 32);
        const float inv = 1.0f / l;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][qt][r] *= DROP ? inv * am_keep(keep, kt, qt, r, p.drop.scale) : inv;
        if (p.lse && lane < 32 && query < T) p.lse[(n * p.H + h) * T + query] = m + logf(l);
    }
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        vc_wave_barrier();
        am_stage(tiles[0], (const vc_bf16*)p.v + rowq * p.ldv + hd + c * AM_D, p.ldv, T, lane);
        vc_wave_barrier();
        vc_f32x16 o[2][2];
        am_zero(o);
        am_mm_tok(o, st, tiles[0], lane);    // O[query][d] = sum_key P[query][key] V[key][d]
        am_store((vc_bf16*)p.o + rowq * p.ldo + hd + c * AM_D, p.ldo, o, T, lane, 1.0f);
    }
}

// r02 forward: ONE WAVE PER HEAD-DIM CHUNK (NCH waves per (clip, head)).  The one-wave kernel above walks the chunks through the same two
// LDS tiles — 2 NCH dependent load -> MFMA phases, ~22 us for a launch of only B x H workgroups.  Here every wave stages its own Q / K / V
// chunk at once (one memory round trip), computes its partial score grid, the partials are summed through LDS in a fixed order (every
// wave ends up with the same bits), each wave runs the (cheap) softmax redundantly and produces its own 64 output columns.
constexpr int AM_CW_WAVE_ELEMS = 3 * AM_T * AM_S;                                    // Q, K, V chunk tiles of one wave
constexpr size_t am_cw_lds_bytes(int nch) { return (size_t)nch * AM_CW_WAVE_ELEMS * 2; }
template <bool DROP, int NCH>
VC_KERNEL __launch_bounds__(64 * NCH) void attn_dec_fwd_cw_kernel(AttnParams p) {
    VC_DYN_SHARED(vc_bf16, lds);
    const int tid = threadIdx.x, lane = tid & 63, c = vc_uniform(tid >> 6);
    vc_bf16* Qs = lds + c * AM_CW_WAVE_ELEMS; vc_bf16* Ks = Qs + AM_T * AM_S; vc_bf16* Vs = Ks + AM_T * AM_S;
    const int h = blockIdx.x % p.H; const long n = blockIdx.x / p.H;
    const int T = p.Tq;
    const long rowq = n * T;
    const int hd = h * AM_D * NCH + c * AM_D;
    am_stage(Qs, (const vc_bf16*)p.q + rowq * p.ldq + hd, p.ldq, T, lane);
    am_stage(Ks, (const vc_bf16*)p.k + rowq * p.ldk + hd, p.ldk, T, lane);
    am_stage(Vs, (const vc_bf16*)p.v + rowq * p.ldv + hd, p.ldv, T, lane);
    vc_sync();                                // every wave's chunk is in LDS
    // every wave accumulates the FULL score grid over the chunks in the order 0 .. NCH-1 — the order of the backward kernel's recompute, so
    // exp(s - lse) is exactly 1 where a query sees a single key (window = 1: dS = 0 bit for bit, like the reference); the 4x redundant MFMAs
    // run on four different SIMDs and cost no wall time, the memory round trip was the price
    vc_f32x16 st[2][2];                       // S^T: [key tile][query tile], lane column = query
    am_zero(st);
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) am_mm_nt(st, lds + cc * AM_CW_WAVE_ELEMS + AM_T * AM_S, lds + cc * AM_CW_WAVE_ELEMS, lane);
    vc_sync();                                // all Q / K fragments consumed: the own Q tile becomes the store staging
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    uint64_t keep = 0;
    if (DROP) keep = am_keep_bits<true>(p.drop, dbase0, T, lane);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int query = qt * 32 + (lane & 31);
        const int qm = query < T ? query : T - 1;              // padding columns: keep the row finite (never stored)
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + am_row(r, lane);
                const float sc = am_visible(qm, key, T, p.window) ? st[kt][qt][r] * p.scale : -INFINITY;
                st[kt][qt][r] = sc; m = fmaxf(m, sc);
            }
        m = fmaxf(m, vc_shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = vc_expf_fast(st[kt][qt][r] - m); st[kt][qt][r] = e; l += e; }
        l += vc_shfl_xor(l, 32);
        const float inv = 1.0f / l;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][qt][r] *= DROP ? inv * am_keep(keep, kt, qt, r, p.drop.scale) : inv;
        if (c == 0 && p.lse && lane < 32 && query < T) p.lse[(n * p.H + h) * T + query] = m + logf(l);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const vc_f32x16 W[2] = {st[0][qt], st[1][qt]};
        vc_f32x16 o[2];
        am_zero1(o);
        am_mm_tok1<false>(o, W, Vs, lane, 0u, 1.0f);           // O[query][d] = sum_key P[query][key] V[key][d], this wave's 64 columns
        am_store_rows(Qs, (vc_bf16*)p.o + rowq * p.ldo + hd, p.ldo, o, qt, T, lane, 1.0f);
    }
}

// Backward: 2 waves per (clip, head) as in the ViT kernel — wave 0 owns "lane = query" (D_i, dQ), wave 1 "lane = key" (dV, dK).
// Phase 1 accumulates both score-shaped grids of each wave over the head-dim chunks, phase 2 re-stages the chunks and emits
// the three gradients chunk by chunk.
template <bool DROP, int NCH>
VC_KERNEL __launch_bounds__(128, 1) void attn_dec_bwd_mfma_kernel(AttnParams p) {
    VC_SHARED __attribute__((aligned(16))) vc_bf16 tiles[4][AM_T * AM_S];     // Q, K, V, dO chunk
    VC_SHARED float lse_s[AM_T];
    VC_SHARED float del_s[AM_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x % p.H; const long n = blockIdx.x / p.H;
    const int T = p.Tq;
    const long rowq = n * T;
    const int hd = h * AM_D * NCH;
    const vc_bf16 *Qs = tiles[0], *Ks = tiles[1], *Vs = tiles[2], *dOs = tiles[3];
    if (tid < AM_T) lse_s[tid] = (tid < T) ? p.lse[(n * p.H + h) * T + tid] : 0.f;
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;

    vc_f32x16 sg[2][2], dg[2][2];            // wave 0: S^T, dP^T [key tile][query tile]; wave 1: S, dP' [query tile][key tile]
    am_zero(sg); am_zero(dg);
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        vc_sync();
        am_stage_nt<128>(tiles[0], (const vc_bf16*)p.q + rowq * p.ldq + hd + c * AM_D, p.ldq, T, tid);
        am_stage_nt<128>(tiles[1], (const vc_bf16*)p.k + rowq * p.ldk + hd + c * AM_D, p.ldk, T, tid);
        am_stage_nt<128>(tiles[2], (const vc_bf16*)p.v + rowq * p.ldv + hd + c * AM_D, p.ldv, T, tid);
        am_stage_nt<128>(tiles[3], (const vc_bf16*)p.dout + rowq * p.lddo + hd + c * AM_D, p.lddo, T, tid);
        vc_sync();
        if (wave == 0) { am_mm_nt(sg, Ks, Qs, lane); am_mm_nt(dg, Vs, dOs, lane); }
        else           { am_mm_nt(sg, Qs, Ks, lane); am_mm_nt(dg, dOs, Vs, lane); }
    }
    uint64_t keep = 0;
    if (wave == 0) {   // ---------------- lane = query: P, D_i, dS^T
        if (DROP) keep = am_keep_bits<true>(p.drop, dbase0, T, lane);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int query = qt * 32 + (lane & 31);
            const float lse = lse_s[query];
            float dsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + am_row(r, lane);
                    const bool ok = query < T && am_visible(query, key, T, p.window);
                    const float pr = ok ? vc_expf_fast(sg[kt][qt][r] * p.scale - lse) : 0.f;
                    if (DROP) dg[kt][qt][r] *= am_keep(keep, kt, qt, r, p.drop.scale);                    // dP = dP' * mask
                    sg[kt][qt][r] = pr; dsum += pr * dg[kt][qt][r];
                }
            dsum += vc_shfl_xor(dsum, 32);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sg[kt][qt][r] = sg[kt][qt][r] * (dg[kt][qt][r] - dsum);   // dS^T (scale folded into the store)
            if (lane < 32) del_s[query] = dsum;
        }
    } else {           // ---------------- lane = key: P (kept in sg), later dS (in dg)
        if (DROP) keep = am_keep_bits<false>(p.drop, dbase0, T, lane);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = kt * 32 + (lane & 31);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int query = qt * 32 + am_row(r, lane);
                    sg[qt][kt][r] = (query < T && am_visible(query, key, T, p.window)) ? vc_expf_fast(sg[qt][kt][r] * p.scale - lse_s[query]) : 0.f;   // P
                }
        }
    }
    vc_sync();                               // D_i published
    if (wave == 1) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int query = qt * 32 + am_row(r, lane);
                    const float ms = DROP ? am_keep(keep, qt, kt, r, p.drop.scale) : 1.0f;
                    dg[qt][kt][r] = sg[qt][kt][r] * (dg[qt][kt][r] * ms - del_s[query]);                          // dS
                }
    }
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        vc_sync();
        am_stage_nt<128>(tiles[0], (const vc_bf16*)p.q + rowq * p.ldq + hd + c * AM_D, p.ldq, T, tid);
        am_stage_nt<128>(tiles[1], (const vc_bf16*)p.k + rowq * p.ldk + hd + c * AM_D, p.ldk, T, tid);
        am_stage_nt<128>(tiles[3], (const vc_bf16*)p.dout + rowq * p.lddo + hd + c * AM_D, p.lddo, T, tid);
        vc_sync();
        vc_f32x16 acc[2][2];
        if (wave == 0) {
            am_zero(acc);
            am_mm_tok(acc, sg, Ks, lane);        // dQ[query][d] = sum_key dS[query][key] K[key][d]
            am_store((vc_bf16*)p.dq + rowq * p.lddq + hd + c * AM_D, p.lddq, acc, T, lane, p.scale);
        } else {
            am_zero(acc);
            am_mm_tok_keep<DROP>(acc, sg, dOs, lane, keep, p.drop.scale);     // dV[key][d] = sum_query P'[query][key] dO[query][d]
            am_store((vc_bf16*)p.dv + rowq * p.lddv + hd + c * AM_D, p.lddv, acc, T, lane, 1.0f);
            am_zero(acc);
            am_mm_tok(acc, dg, Qs, lane);        // dK[key][d] = sum_query dS[query][key] Q[query][d]
            am_store((vc_bf16*)p.dk + rowq * p.lddk + hd + c * AM_D, p.lddk, acc, T, lane, p.scale);
        }
    }
}

// r02 backward: FOUR waves per (clip, head) — (orientation) x (half of the head-dim chunks), one per SIMD with the full register file (an
// eight-wave form, one per chunk, spilled 120 registers at the 256-register cap).  All 4 NCH chunk tiles (Q, K, V, dO) are staged at once
// (one memory round trip instead of 2 NCH); every wave of an orientation accumulates the two score-shaped grids over the chunks in the order
// 0 .. NCH-1 (bit-identical to the forward's lse, see above; redundant MFMAs on otherwise idle SIMDs), then produces the gradients of its own
// chunks: dQ_c (query waves), dV_c and dK_c (key waves) — token-contraction products with swapped operands and whole-row stores through a
// wave-private staging area carved out of the V tiles, which are dead after the first phase.
constexpr size_t am_cwb_lds_bytes(int nch) { return (size_t)4 * nch * AM_T * AM_S * 2 + 2 * AM_T * sizeof(float); }
template <bool DROP>
VC_DEV void am_mm_tok1k(vc_f32x16 (&out)[2], const vc_f32x16 (&W)[2], const vc_bf16* Y, int lane, uint64_t keep, int tj, float scale) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const vc_s16x8 b = DROP ? am_pack_keep(W[tt], s, keep, tt, tj, scale) : am_pack(W[tt], s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) out[dt] = vc_mfma_32x32x16_bf16(am_frag_tr(Y, tt * 32 + 16 * s, dt * 32, lane), b, out[dt]);
        }
}
template <bool DROP, int NCH>
VC_KERNEL __launch_bounds__(256, 1) void attn_dec_bwd_cw_kernel(AttnParams p) {
    VC_DYN_SHARED(vc_bf16, lds);
    constexpr int TILE = AM_T * AM_S, CPW = NCH / 2;             // chunks per wave in the second phase
    float* lse_s = reinterpret_cast<float*>(lds + 4 * NCH * TILE);
    float* del_s = lse_s + AM_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = vc_uniform(tid >> 6);
    const int o = wave >> 1, half = wave & 1;                    // orientation (0 = lane is a query, 1 = lane is a key), which half of the chunks
    const int h = blockIdx.x % p.H; const long n = blockIdx.x / p.H;
    const int T = p.Tq;
    const long rowq = n * T;
    const int hd = h * AM_D * NCH;
    auto Qt = [&](int cc) { return lds + (0 * NCH + cc) * TILE; };
    auto Kt = [&](int cc) { return lds + (1 * NCH + cc) * TILE; };
    auto Vt = [&](int cc) { return lds + (2 * NCH + cc) * TILE; };
    auto Ot = [&](int cc) { return lds + (3 * NCH + cc) * TILE; };
    {   // wave w stages all NCH chunks of tensor w (Q, K, V, dO): every load is issued before the first LDS store (one memory round trip)
        const vc_bf16* g = wave == 0 ? (const vc_bf16*)p.q : (wave == 1 ? (const vc_bf16*)p.k : (wave == 2 ? (const vc_bf16*)p.v : (const vc_bf16*)p.dout));
        const long ld = wave == 0 ? p.ldq : (wave == 1 ? p.ldk : (wave == 2 ? p.ldv : p.lddo));
        g += rowq * ld + hd;
        vc_u32x4 v[NCH][AM_T / 8];
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
            for (int it = 0; it < AM_T / 8; ++it) {
                const int row = it * 8 + (lane >> 3), col = (lane & 7) * 8;
                v[cc][it] = *reinterpret_cast<const vc_u32x4*>(g + (long)(row < T ? row : T - 1) * ld + cc * AM_D + col);
            }
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
            for (int it = 0; it < AM_T / 8; ++it) {
                const int row = it * 8 + (lane >> 3), col = (lane & 7) * 8;
                vc_u32x4 w = v[cc][it];
                if (row >= T) { w.x = 0u; w.y = 0u; w.z = 0u; w.w = 0u; }
                *reinterpret_cast<vc_u32x4*>(lds + (wave * NCH + cc) * TILE + row * AM_S + col) = w;
            }
    }
    if (tid < AM_T) lse_s[tid] = (tid < T) ? p.lse[(n * p.H + h) * T + tid] : 0.f;
    vc_sync();
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    vc_f32x16 sg[2][2], dg[2][2];            // query waves: S^T, dP^T [key tile][query tile]; key waves: S, dP' [query tile][key tile]
    am_zero(sg); am_zero(dg);
#pragma unroll 1
    for (int cc = 0; cc < NCH; ++cc) {
        if (o == 0) { am_mm_nt(sg, Kt(cc), Qt(cc), lane); am_mm_nt(dg, Vt(cc), Ot(cc), lane); }
        else        { am_mm_nt(sg, Qt(cc), Kt(cc), lane); am_mm_nt(dg, Ot(cc), Vt(cc), lane); }
    }
    uint64_t keep = 0;
    if (o == 0) {      // ---------------- lane = query: P, D_i, dS^T
        if (DROP) keep = am_keep_bits<true>(p.drop, dbase0, T, lane);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int query = qt * 32 + (lane & 31);
            const float lse = lse_s[query];
            float dsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + am_row(r, lane);
                    const bool ok = query < T && am_visible(query, key, T, p.window);
                    const float pr = ok ? vc_expf_fast(sg[kt][qt][r] * p.scale - lse) : 0.f;
                    if (DROP) dg[kt][qt][r] *= am_keep(keep, kt, qt, r, p.drop.scale);                    // dP = dP' * mask
                    sg[kt][qt][r] = pr; dsum += pr * dg[kt][qt][r];
                }
            dsum += vc_shfl_xor(dsum, 32);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sg[kt][qt][r] = sg[kt][qt][r] * (dg[kt][qt][r] - dsum);   // dS^T (scale folded into the store)
            if (half == 0 && lane < 32) del_s[query] = dsum;
        }
    } else {           // ---------------- lane = key: P (kept in sg), later dS (in dg)
        if (DROP) keep = am_keep_bits<false>(p.drop, dbase0, T, lane);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = kt * 32 + (lane & 31);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int query = qt * 32 + am_row(r, lane);
                    sg[qt][kt][r] = (query < T && am_visible(query, key, T, p.window)) ? vc_expf_fast(sg[qt][kt][r] * p.scale - lse_s[query]) : 0.f;   // P
                }
        }
    }
    vc_sync();                               // D_i published; nobody reads V any more: its tiles become the store staging
    vc_bf16* stage = Vt(wave >> 1) + (wave & 1) * 32 * AM_S;
    if (o == 0) {
#pragma unroll 1
      for (int c = half * CPW; c < (half + 1) * CPW; ++c)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const vc_f32x16 W[2] = {sg[0][qt], sg[1][qt]};
            vc_f32x16 acc[2];
            am_zero1(acc);
            am_mm_tok1k<false>(acc, W, Kt(c), lane, 0, 0, 1.0f);             // dQ[query][d] = sum_key dS[query][key] K[key][d]
            am_store_rows(stage, (vc_bf16*)p.dq + rowq * p.lddq + hd + c * AM_D, p.lddq, acc, qt, T, lane, p.scale);
        }
    } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int query = qt * 32 + am_row(r, lane);
                    const float ms = DROP ? am_keep(keep, qt, kt, r, p.drop.scale) : 1.0f;
                    dg[qt][kt][r] = sg[qt][kt][r] * (dg[qt][kt][r] * ms - del_s[query]);                          // dS
                }
#pragma unroll 1
      for (int c = half * CPW; c < (half + 1) * CPW; ++c)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            vc_f32x16 acc[2];
            { const vc_f32x16 W[2] = {sg[0][kt], sg[1][kt]};
              am_zero1(acc);
              am_mm_tok1k<DROP>(acc, W, Ot(c), lane, keep, kt, p.drop.scale);   // dV[key][d] = sum_query P'[query][key] dO[query][d]
              am_store_rows(stage, (vc_bf16*)p.dv + rowq * p.lddv + hd + c * AM_D, p.lddv, acc, kt, T, lane, 1.0f); }
            { const vc_f32x16 W[2] = {dg[0][kt], dg[1][kt]};
              am_zero1(acc);
              am_mm_tok1k<false>(acc, W, Qt(c), lane, 0, 0, 1.0f);               // dK[key][d] = sum_query dS[query][key] Q[query][d]
              am_store_rows(stage, (vc_bf16*)p.dk + rowq * p.lddk + hd + c * AM_D, p.lddk, acc, kt, T, lane, p.scale); }
        }
    }
}

// ====================================================================================================================
// Decoder attention for T > 64 (any horizon; the canonical maximum is 186 = three 64-step blocks, the reference allows max_ep_len = 1000):
// the same machinery per (query block, key block) pair, STREAMED over the blocks of the other side.
//
// r04 form.  r01-r03 ran ONE wave per (clip, head, 64-row block) that walked the four 64-wide head-dim chunks of every visible block pair through
// two small LDS tiles: 192 single-wave workgroups on 1 024 SIMDs, each a chain of ~30-40 dependent global-load -> LDS -> MFMA phases
// (109 us per backward-dQ launch for 2.3 GFLOP at T = 186; three key blocks at most because the score grids of all of them sat in registers).
// Here a workgroup is NCH waves (one per head-dim chunk, as in the T <= 64 `cw` kernels above): per block pair ONE memory round trip stages every
// chunk, every wave accumulates the full score grids over the chunks in the order 0 .. NCH-1 (redundant MFMAs on four different SIMDs; the same
// order in all three kernels, so exp(s - lse) is exactly 1 where a query sees a single key and dS = 0 bit for bit, as in the reference) and owns
// its chunk of the outputs.
//   forward  : NCH waves per (clip, head, query block); key blocks visited in ascending order with a running (max, sum) per query
//              (online softmax: P = exp(s - running max) feeds PV unnormalised, O is rescaled when the max moves and divided by the sum at the end)
//   backward : attn_dec_bwd_q (lane = query) accumulates D_i = sum_key P dP over its key blocks in a first sweep with exactly the P and dP the second
//              sweep uses (the rowsum(dO * O) identity would leave rounding noise where a query sees one key), then dQ_c += dS K_c per key block
//              straight from the accumulator registers; attn_dec_bwd_kv (lane = key, launched after it, reads D_i) sweeps its query blocks and
//              accumulates dV_c += P'^T dO_c, dK_c += dS^T Q_c.
// ====================================================================================================================
constexpr size_t am_blk_fwd_lds_bytes(int nch) { return (size_t)3 * nch * AM_T * AM_S * 2; }
constexpr size_t am_blk_bwd_lds_bytes(int nch) { return (size_t)4 * nch * AM_T * AM_S * 2 + 2 * AM_T * sizeof(float); }

// first / last key block a query block can see, and vice versa (causal + window band)
VC_DEV int am_kb_lo(int qb, int window) { const int k = qb * AM_T - window + 1; return k > 0 ? k / AM_T : 0; }
VC_DEV int am_qb_hi(int kb, int window, int nblk) { const int q = (kb * AM_T + AM_T - 1 + window - 1) / AM_T; return q < nblk - 1 ? q : nblk - 1; }

// two [T x 64] head slices into two wave-private LDS tiles, all sixteen 16-byte loads in flight before the first LDS store (am_stage twice
// would make the second tensor's loads wait behind the first one's stores)
VC_DEV void am_stage2(vc_bf16* tile_a, const vc_bf16* ga, long lda, vc_bf16* tile_b, const vc_bf16* gb, long ldb, int T, int lane) {
    vc_u32x4 va[AM_T / 8], vb[AM_T / 8];
#pragma unroll
    for (int it = 0; it < AM_T / 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        const int rr = row < T ? row : T - 1;
        va[it] = *reinterpret_cast<const vc_u32x4*>(ga + (long)rr * lda + c);
        vb[it] = *reinterpret_cast<const vc_u32x4*>(gb + (long)rr * ldb + c);
    }
#pragma unroll
    for (int it = 0; it < AM_T / 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        vc_u32x4 wa = va[it], wb = vb[it];
        if (row >= T) { wa.x = wa.y = wa.z = wa.w = 0u; wb.x = wb.y = wb.z = wb.w = 0u; }
        *reinterpret_cast<vc_u32x4*>(tile_a + row * AM_S + c) = wa;
        *reinterpret_cast<vc_u32x4*>(tile_b + row * AM_S + c) = wb;
    }
}

template <bool DROP, int NCH>
VC_KERNEL __launch_bounds__(64 * NCH, 1) void attn_dec_fwd_blk_kernel(AttnParams p) {
    VC_DYN_SHARED(vc_bf16, lds);
    constexpr int TILE = AM_T * AM_S;
    const int tid = threadIdx.x, lane = tid & 63, c = vc_uniform(tid >> 6);
    auto Qt = [&](int cc) { return lds + cc * TILE; };
    auto Kt = [&](int cc) { return lds + (NCH + cc) * TILE; };
    auto Vt = [&](int cc) { return lds + (2 * NCH + cc) * TILE; };
    const int T = p.Tq, nblk = (T + AM_T - 1) / AM_T;
    const int qb = blockIdx.x % nblk; const long bh = blockIdx.x / nblk;
    const int h = (int)(bh % p.H); const long n = bh / p.H;
    const long rowq = n * T;
    const int hd = h * AM_D * NCH + c * AM_D;                                 // this wave's head-dim chunk
    const int q0 = qb * AM_T, tq = T - q0 < AM_T ? T - q0 : AM_T;             // rows of this query block
    const int kb0 = am_kb_lo(qb, p.window);
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    am_stage(Qt(c), (const vc_bf16*)p.q + (rowq + q0) * p.ldq + hd, p.ldq, tq, lane);
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};           // running max / sum of this lane's two query columns
    vc_f32x16 o[2][2];                                                        // [query tile][d tile]: registers walk the head dim, lane = query
    am_zero(o);
#pragma unroll 1
    for (int kb = kb0; kb <= qb; ++kb) {
        const int k0 = kb * AM_T, tk = T - k0 < AM_T ? T - k0 : AM_T;
        if (kb != kb0) vc_sync();                                             // the previous block's K / V fragments are consumed
        am_stage2(Kt(c), (const vc_bf16*)p.k + (rowq + k0) * p.ldk + hd, p.ldk, Vt(c), (const vc_bf16*)p.v + (rowq + k0) * p.ldv + hd, p.ldv, tk, lane);
        vc_sync();
        vc_f32x16 st[2][2];                                                   // S^T: [key tile][query tile], lane column = query
        am_zero(st);
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) am_mm_nt(st, Kt(cc), Qt(cc), lane);
        uint64_t keep = 0;
        if (DROP) keep = am_keep_bits_g<true>(p.drop, dbase0, T, q0, k0, lane);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int ql = qt * 32 + (lane & 31);
            const int query = q0 + (ql < tq ? ql : tq - 1);                   // padding columns: keep the row finite (never stored)
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + am_row(r, lane);
                    const float sc = am_visible(query, key, T, p.window) ? st[kt][qt][r] * p.scale : -INFINITY;
                    st[kt][qt][r] = sc; m = fmaxf(m, sc);
                }
            m = fmaxf(m, vc_shfl_xor(m, 32));
            const float m_new = fmaxf(m_run[qt], m);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;             // a query that sees no key of the blocks so far: everything stays 0
            const float alpha = vc_expf_fast(m_run[qt] - m_use);
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float e = vc_expf_fast(st[kt][qt][r] - m_use); st[kt][qt][r] = e; l += e; }
            l += vc_shfl_xor(l, 32);
            l_run[qt] = l_run[qt] * alpha + l; m_run[qt] = m_new;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
            const vc_f32x16 W[2] = {st[0][qt], st[1][qt]};
            am_mm_tok1k<DROP>(o[qt], W, Vt(c), lane, keep, qt, p.drop.scale);      // O[query][d] += sum_key P'[query][key] V[key][d], this wave's 64 columns
        }
    }
    vc_sync();                                                                // every wave is done with the Q tiles: the own one becomes the store staging
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int ql = qt * 32 + (lane & 31);
        am_store_rows(Qt(c), (vc_bf16*)p.o + (rowq + q0) * p.ldo + hd, p.ldo, o[qt], qt, tq, lane, 1.0f / l_run[qt]);
        if (c == 0 && p.lse && lane < 32 && ql < tq) p.lse[(n * p.H + h) * T + q0 + ql] = m_run[qt] + logf(l_run[qt]);
    }
}

// backward, lane = query: D_i and dQ of one query block (D_i is also written to p.delta for the dK/dV kernel that is launched next)
template <bool DROP, int NCH>
VC_KERNEL __launch_bounds__(64 * NCH, 1) void attn_dec_bwd_q_blk_kernel(AttnParams p) {
    VC_DYN_SHARED(vc_bf16, lds);
    constexpr int TILE = AM_T * AM_S;
    const int tid = threadIdx.x, lane = tid & 63, c = vc_uniform(tid >> 6);
    auto Qt = [&](int cc) { return lds + cc * TILE; };
    auto Ot = [&](int cc) { return lds + (NCH + cc) * TILE; };
    auto Kt = [&](int cc) { return lds + (2 * NCH + cc) * TILE; };
    auto Vt = [&](int cc) { return lds + (3 * NCH + cc) * TILE; };
    const int T = p.Tq, nblk = (T + AM_T - 1) / AM_T;
    const int qb = blockIdx.x % nblk; const long bh = blockIdx.x / nblk;
    const int h = (int)(bh % p.H); const long n = bh / p.H;
    const long rowq = n * T;
    const int hd = h * AM_D * NCH + c * AM_D;
    const int q0 = qb * AM_T, tq = T - q0 < AM_T ? T - q0 : AM_T;
    const int kb0 = am_kb_lo(qb, p.window);
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    am_stage2(Qt(c), (const vc_bf16*)p.q + (rowq + q0) * p.ldq + hd, p.ldq, Ot(c), (const vc_bf16*)p.dout + (rowq + q0) * p.lddo + hd, p.lddo, tq, lane);
    float ls[2], dsum[2] = {0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) { const int ql = qt * 32 + (lane & 31); ls[qt] = ql < tq ? p.lse[(n * p.H + h) * T + q0 + ql] : 0.f; }
    vc_f32x16 dq[2][2];                                                       // [query tile][d tile]: registers walk the head dim, lane = query
    am_zero(dq);
    bool first = true;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll 1
        for (int kb = kb0; kb <= qb; ++kb) {
            const int k0 = kb * AM_T, tk = T - k0 < AM_T ? T - k0 : AM_T;
            if (!first) vc_sync();                                            // the previous block's K / V fragments are consumed
            first = false;
            am_stage2(Kt(c), (const vc_bf16*)p.k + (rowq + k0) * p.ldk + hd, p.ldk, Vt(c), (const vc_bf16*)p.v + (rowq + k0) * p.ldv + hd, p.ldv, tk, lane);
            vc_sync();
            vc_f32x16 sg[2][2], dg[2][2];                                     // S^T, dP'^T: [key tile][query tile]
            am_zero(sg); am_zero(dg);
#pragma unroll 1
            for (int cc = 0; cc < NCH; ++cc) { am_mm_nt(sg, Kt(cc), Qt(cc), lane); am_mm_nt(dg, Vt(cc), Ot(cc), lane); }
            uint64_t keep = 0;
            if (DROP) keep = am_keep_bits_g<true>(p.drop, dbase0, T, q0, k0, lane);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int ql = qt * 32 + (lane & 31);
                const bool qok = ql < tq;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + kt * 32 + am_row(r, lane);
                        const bool ok = qok && am_visible(q0 + ql, key, T, p.window);
                        const float pr = ok ? vc_expf_fast(sg[kt][qt][r] * p.scale - ls[qt]) : 0.f;
                        const float dpm = dg[kt][qt][r] * (DROP ? am_keep(keep, kt, qt, r, p.drop.scale) : 1.0f);      // dP = dP' * mask
                        if (pass == 0) dsum[qt] += pr * dpm;
                        else sg[kt][qt][r] = pr * (dpm - dsum[qt]);                                                       // dS^T (scale folded into the store)
                    }
                if (pass == 1) {
                    const vc_f32x16 W[2] = {sg[0][qt], sg[1][qt]};
                    am_mm_tok1k<false>(dq[qt], W, Kt(c), lane, 0, 0, 1.0f);   // dQ[query][d] += sum_key dS[query][key] K[key][d], this wave's 64 columns
                }
            }
        }
        if (pass == 0) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                dsum[qt] += vc_shfl_xor(dsum[qt], 32);
                const int ql = qt * 32 + (lane & 31);
                if (c == 0 && lane < 32 && ql < tq) p.delta[(n * p.H + h) * T + q0 + ql] = dsum[qt];
            }
        }
    }
    vc_sync();                                                                // Q tiles free: the own one becomes the store staging
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) am_store_rows(Qt(c), (vc_bf16*)p.dq + (rowq + q0) * p.lddq + hd, p.lddq, dq[qt], qt, tq, lane, p.scale);
}

// backward, lane = key: dK, dV of one key block
template <bool DROP, int NCH>
VC_KERNEL __launch_bounds__(64 * NCH, 1) void attn_dec_bwd_kv_blk_kernel(AttnParams p) {
    VC_DYN_SHARED(vc_bf16, lds);
    constexpr int TILE = AM_T * AM_S;
    float* lse_s = reinterpret_cast<float*>(lds + 4 * NCH * TILE);
    float* del_s = lse_s + AM_T;
    const int tid = threadIdx.x, lane = tid & 63, c = vc_uniform(tid >> 6);
    auto Kt = [&](int cc) { return lds + cc * TILE; };
    auto Vt = [&](int cc) { return lds + (NCH + cc) * TILE; };
    auto Qt = [&](int cc) { return lds + (2 * NCH + cc) * TILE; };
    auto Ot = [&](int cc) { return lds + (3 * NCH + cc) * TILE; };
    const int T = p.Tq, nblk = (T + AM_T - 1) / AM_T;
    const int kb = blockIdx.x % nblk; const long bh = blockIdx.x / nblk;
    const int h = (int)(bh % p.H); const long n = bh / p.H;
    const long rowq = n * T;
    const int hd = h * AM_D * NCH + c * AM_D;
    const int k0 = kb * AM_T, tk = T - k0 < AM_T ? T - k0 : AM_T;
    const int qb1 = am_qb_hi(kb, p.window, nblk);
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    am_stage2(Kt(c), (const vc_bf16*)p.k + (rowq + k0) * p.ldk + hd, p.ldk, Vt(c), (const vc_bf16*)p.v + (rowq + k0) * p.ldv + hd, p.ldv, tk, lane);
    vc_f32x16 dv[2][2], dk[2][2];                                             // [key tile][d tile]: registers walk the head dim, lane = key
    am_zero(dv); am_zero(dk);
#pragma unroll 1
    for (int qb = kb; qb <= qb1; ++qb) {
        const int q0 = qb * AM_T, tq = T - q0 < AM_T ? T - q0 : AM_T;
        if (qb != kb) vc_sync();                                              // the previous block's Q / dO fragments and statistics are consumed
        am_stage2(Qt(c), (const vc_bf16*)p.q + (rowq + q0) * p.ldq + hd, p.ldq, Ot(c), (const vc_bf16*)p.dout + (rowq + q0) * p.lddo + hd, p.lddo, tq, lane);
        if (tid < AM_T) {
            lse_s[tid] = tid < tq ? p.lse[(n * p.H + h) * T + q0 + tid] : 0.f;
            del_s[tid] = tid < tq ? p.delta[(n * p.H + h) * T + q0 + tid] : 0.f;
        }
        vc_sync();
        uint64_t keep = 0;
        if (DROP) keep = am_keep_bits_g<false>(p.drop, dbase0, T, q0, k0, lane);
        // one 32-key column tile at a time (score-shaped grids of 2 x 1 tiles: with the four 2 x 2 output grids that stay live across the sweep a
        // 2 x 2 score pair would not fit the register file)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            vc_f32x16 sg[2], dg[2];                                           // S, dP': [query tile], lane column = key of tile kt
            am_zero1(sg); am_zero1(dg);
#pragma unroll 1
            for (int cc = 0; cc < NCH; ++cc) { am_mm_nt1(sg, Qt(cc), Kt(cc), kt, lane); am_mm_nt1(dg, Ot(cc), Vt(cc), kt, lane); }
            const int key = k0 + kt * 32 + (lane & 31);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = qt * 32 + am_row(r, lane);
                    const bool ok = ql < tq && am_visible(q0 + ql, key, T, p.window);
                    const float pr = ok ? vc_expf_fast(sg[qt][r] * p.scale - lse_s[ql]) : 0.f;
                    const float ms = DROP ? am_keep(keep, qt, kt, r, p.drop.scale) : 1.0f;
                    dg[qt][r] = pr * (dg[qt][r] * ms - del_s[ql]);                                                        // dS
                    sg[qt][r] = pr;                                                                                       // P (the mask is applied while packing)
                }
            am_mm_tok1k<DROP>(dv[kt], sg, Ot(c), lane, keep, kt, p.drop.scale);       // dV[key][d] += sum_query P'[query][key] dO[query][d]
            am_mm_tok1k<false>(dk[kt], dg, Qt(c), lane, 0, 0, 1.0f);                  // dK[key][d] += sum_query dS[query][key] Q[query][d]
        }
    }
    vc_sync();                                                                // Q tiles free: the own one becomes the store staging
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        am_store_rows(Qt(c), (vc_bf16*)p.dv + (rowq + k0) * p.lddv + hd, p.lddv, dv[kt], kt, tk, lane, 1.0f);
        am_store_rows(Qt(c), (vc_bf16*)p.dk + (rowq + k0) * p.lddk + hd, p.lddk, dk[kt], kt, tk, lane, p.scale);
    }
}
