// loss.h — fused loss + metrics + dlogits for MultiClassesTrainer.compute_loss
// (reference trainer.py:853-917 flexible_cross_entropy, :935-1063 compute_loss).
//
// HBM-bound: 24 KB of fp32 logits per token are read once for the loss/argmax pass and once more for the
// dlogits pass (the masked means need the global row counts first).  No host syncs: the ~40 `.item()`
// counters of the reference are produced by one finalize block and copied back in a single D2H.
//
//  pass 1  loss_rows_kernel    : one wave per (row m, param head i): max / first-argmax / LSE via wave
//                                shuffles, window sum, per-row numerator & denominator; cmd head per thread
//  pass 2  loss_finalize_kernel: one block: masked means -> loss, per-head scales, metric counters
//  pass 3  loss_dlogits_kernel : d logits = scale * (softmax - soft_target), zero for excluded rows
#pragma once
#include "vc_rt.h"

enum { VC_NPARAM = 6, VC_NCMD = 5, VC_NVAL = 1000, VC_NMETRIC = 32 };
// metric slots
enum { MET_CMD_CORRECT = 0 /*5*/, MET_CMD_COUNT = 5 /*5*/, MET_PAR_CORRECT = 10 /*6*/, MET_PAR_COUNT = 16 /*6*/,
       MET_CMD_CORRECT_TOPK = 22, MET_CMD_COUNT_TOPK = 23, MET_PAR_CORRECT_TOPK = 24, MET_PAR_COUNT_TOPK = 25,
       MET_CORRECT = 26, MET_TOTAL = 27 };

struct LossParams {
    const float* cmds;  long ldc;       // [M, 5]
    const float* pars;  long ldp;       // [M, 6*1000]
    const float* targets;               // [M, 7] raw actions[:, 1:] (float, -1 = ignore)
    long M; int T;                      // T = steps per clip (row m -> t = m % T), for the [:, :30] counters
    int use_mse;                        // 1: flexible CE (window), 0: class-weighted CE
    int tol[VC_NPARAM]; int above[VC_NPARAM];
    float label_w[VC_NCMD];             // class_weights.json "Label"
    int param_to_label[VC_NPARAM];
    const float* class_w;               // [6][1000] per-class weights (use_mse = 0), may be null
    // per-row scratch
    float* row_num; float* row_den; float* row_lse; int* row_arg;   // [M*7]: heads 0..5 = params, 6 = cmd
    // outputs
    float* loss_out;                    // [8]: total, cmd, p0..p5
    float* scales;                      // [7]: per-head dlogit scale (params 0..5, cmd)
    int* metrics;                       // [VC_NMETRIC]
    float* dcmds; long lddc; float* dpars; long lddp;
    float gmul;                         // dlogits are written times this (fp16 engines in deferred-unscale mode: the gradient scale, a power of two; else 1)
};

VC_KERNEL __launch_bounds__(256) void loss_rows_kernel(LossParams p) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= p.M * VC_NPARAM) return;
    const long m = wid / VC_NPARAM; const int i = (int)(wid % VC_NPARAM);
    const float* z = p.pars + m * p.ldp + (long)i * VC_NVAL;
    const int t = (int)p.targets[m * 7 + 1 + i];
    // first-index argmax + max
    float best = -INFINITY; int bidx = 0x7fffffff;
    for (int j = lane; j < VC_NVAL; j += 64) { float v = z[j]; if (v > best) { best = v; bidx = j; } }
    for (int s = 32; s >= 1; s >>= 1) {
        float ov = vc_shfl_xor(best, s); int oi = vc_shfl_xor(bidx, s);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    float se = 0.f, ws = 0.f;
    const int whi = (t + p.tol[i] - 1 < VC_NVAL - 1) ? (t + p.tol[i] - 1) : (VC_NVAL - 1);
    for (int j = lane; j < VC_NVAL; j += 64) {
        float v = z[j];
        se += expf(v - best);
        if (p.use_mse && j >= t && j <= whi) ws += v;
    }
    se = vc_wave_sum(se); ws = vc_wave_sum(ws);
    const float lse = best + logf(se);
    float num = 0.f, den = 0.f;
    if (t != -1) {
        if (p.use_mse) {
            const bool inside = (bidx >= t) && (bidx <= whi);
            if (!inside) { num = lse - ws / (float)(whi - t + 1); den = 1.f; }
        } else {
            const float w = p.class_w[i * VC_NVAL + t];
            num = w * (lse - z[t]); den = w;
        }
    }
    if (lane == 0) {
        p.row_num[m * 7 + i] = num; p.row_den[m * 7 + i] = den; p.row_lse[m * 7 + i] = lse; p.row_arg[m * 7 + i] = bidx;
    }
}

VC_KERNEL __launch_bounds__(256) void loss_cmd_rows_kernel(LossParams p) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= p.M) return;
    const float* z = p.cmds + m * p.ldc;
    const int t = (int)p.targets[m * 7];
    float best = z[0]; int bidx = 0;
    for (int j = 1; j < VC_NCMD; ++j) if (z[j] > best) { best = z[j]; bidx = j; }
    float se = 0.f;
    for (int j = 0; j < VC_NCMD; ++j) se += expf(z[j] - best);
    const float lse = best + logf(se);
    float num = 0.f, den = 0.f;
    if (t != -1) { const float w = p.label_w[t]; num = w * (lse - z[t]); den = w; }
    p.row_num[m * 7 + 6] = num; p.row_den[m * 7 + 6] = den; p.row_lse[m * 7 + 6] = lse; p.row_arg[m * 7 + 6] = bidx;
}

VC_KERNEL __launch_bounds__(256) void loss_finalize_kernel(LossParams p) {
    VC_SHARED float fs[256][14];
    VC_SHARED int is[256][VC_NMETRIC];
    const int tid = threadIdx.x;
    float num[7], den[7]; int met[VC_NMETRIC];
    for (int h = 0; h < 7; ++h) { num[h] = 0.f; den[h] = 0.f; }
    for (int k = 0; k < VC_NMETRIC; ++k) met[k] = 0;
    for (long m = tid; m < p.M; m += 256) {
        for (int h = 0; h < 7; ++h) { num[h] += p.row_num[m * 7 + h]; den[h] += p.row_den[m * 7 + h]; }
        const int acmd = (int)p.targets[m * 7], cp = p.row_arg[m * 7 + 6];
        const bool topk = (int)(m % p.T) < 30;
        const bool cmask = acmd != -1;
        if (cmask) {
            met[MET_TOTAL]++; if (topk) met[MET_CMD_COUNT_TOPK]++;
            if (acmd >= 0 && acmd < VC_NCMD) { met[MET_CMD_COUNT + acmd]++; if (cp == acmd) met[MET_CMD_CORRECT + acmd]++; }
            if (cp == acmd) { met[MET_CORRECT]++; if (topk) met[MET_CMD_CORRECT_TOPK]++; }
        }
        for (int i = 0; i < VC_NPARAM; ++i) {
            const int a = (int)p.targets[m * 7 + 1 + i];
            if (!(cmask && a != -1)) continue;
            met[MET_PAR_COUNT + i]++; met[MET_TOTAL]++; if (topk) met[MET_PAR_COUNT_TOPK]++;
            if (cp != acmd) continue;
            const int d = p.row_arg[m * 7 + i] - a;
            bool ok;
            if (p.use_mse && p.above[i]) ok = (d >= 0) && (d < p.tol[i]);
            else ok = (d < 3) && (d > -3);
            if (ok) { met[MET_PAR_CORRECT + i]++; met[MET_CORRECT]++; }
            // the reference's non-mse "topk" counter uses exact equality (trainer.py:1015)
            const bool ok_topk = p.use_mse ? ok : (d == 0);
            if (topk && ok_topk) met[MET_PAR_CORRECT_TOPK]++;
        }
    }
    for (int h = 0; h < 7; ++h) { fs[tid][h] = num[h]; fs[tid][7 + h] = den[h]; }
    for (int k = 0; k < VC_NMETRIC; ++k) is[tid][k] = met[k];
    vc_sync();
    if (tid < 14) { float s = 0.f; for (int r = 0; r < 256; ++r) s += fs[r][tid]; fs[0][tid] = s; }
    if (tid >= 64 && tid < 64 + VC_NMETRIC) { int k = tid - 64; int s = 0; for (int r = 0; r < 256; ++r) s += is[r][k]; p.metrics[k] = s; }
    vc_sync();
    if (tid == 0) {
        const float lcmd = fs[0][6] / fs[0][7 + 6];             // 0/0 -> NaN like F.cross_entropy on all-ignored
        float total = 2.0f * lcmd;
        p.loss_out[1] = lcmd;
        p.scales[6] = 2.0f / fs[0][7 + 6];
        for (int i = 0; i < VC_NPARAM; ++i) {
            const float d = fs[0][7 + i];
            const float lw = p.label_w[p.param_to_label[i]];
            float lp = 0.f, sc = 0.f;
            if (d > 0.f) { lp = fs[0][i] / d; sc = lw / d; }   // no rows -> constant 0.0 (trainer.py:895-896); 0/0 NaN skipped (:961)
            p.loss_out[2 + i] = lp; p.scales[i] = sc;
            total += lp * lw;
        }
        p.loss_out[0] = total;
    }
}

VC_KERNEL __launch_bounds__(256) void loss_dlogits_kernel(LossParams p) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= p.M * 7) return;
    const long m = wid / 7; const int h = (int)(wid % 7);
    const float den = p.row_den[m * 7 + h], lse = p.row_lse[m * 7 + h], sc = p.scales[h] * p.gmul;
    if (h == 6) {
        const int t = (int)p.targets[m * 7];
        if (lane < VC_NCMD) {
            float g = 0.f;
            if (t != -1) g = sc * den * (expf(p.cmds[m * p.ldc + lane] - lse) - (lane == t ? 1.f : 0.f));
            p.dcmds[m * p.lddc + lane] = g;
        }
        return;
    }
    const float* z = p.pars + m * p.ldp + (long)h * VC_NVAL;
    float* dz = p.dpars + m * p.lddp + (long)h * VC_NVAL;
    const int t = (int)p.targets[m * 7 + 1 + h];
    const int whi = p.use_mse ? ((t + p.tol[h] - 1 < VC_NVAL - 1) ? (t + p.tol[h] - 1) : (VC_NVAL - 1)) : t;
    const float wsz = 1.0f / (float)(whi - t + 1);
    const float k = sc * den;
    for (int j = lane; j < VC_NVAL; j += 64) {
        float g = 0.f;
        if (den != 0.f) g = k * (expf(z[j] - lse) - ((j >= t && j <= whi) ? wsz : 0.f));
        dz[j] = g;
    }
}
